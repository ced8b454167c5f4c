"""Coordinate helpers with the behaviour of ``sorrel/location.py`` (host side only).

Convention used everywhere in Sorrel and in this engine: a location is
``(y, x, z)`` = (row, column, layer); "up" is ``y - 1`` (``sorrel/agents/agent.py:200-211``).
``Location`` names its first two coordinates ``x`` and ``y`` in storage order, as the
reference does (``sorrel/location.py:21-27``).

>>> Location(1, 2, 3) + Location(2, 4, 8)
Location(3, 6, 11)
>>> Location(2, 4) * 3
Location(6, 12, 0)
"""
from __future__ import annotations

# unit steps (first coord, second coord) for forward/right/backward/left, indexed by the
# compass direction 0 = north/up, 1 = east/right, 2 = south/down, 3 = west/left
_STEPS = (
    ((-1, 0), (0, 1), (1, 0), (0, -1)),
    ((0, 1), (1, 0), (0, -1), (-1, 0)),
    ((1, 0), (0, -1), (-1, 0), (0, 1)),
    ((0, -1), (-1, 0), (0, 1), (1, 0)),
)


class Location(tuple):
    """Immutable 2- or 3-coordinate location supporting ``+`` and scalar ``*``."""

    def __new__(cls, *coords):
        return super().__new__(cls, coords)

    def __init__(self, *coords):
        self.dims = len(coords)
        self.x, self.y = coords[0], coords[1]
        self.z = coords[2] if self.dims > 2 else 0

    def to_tuple(self):
        return (self.x, self.y) if self.dims == 2 else (self.x, self.y, self.z)

    def __repr__(self):
        return f"Location({self.x}, {self.y}, {self.z})"

    __str__ = __repr__

    def __add__(self, other):
        if isinstance(other, Vector):
            other = other.compute()
        if not isinstance(other, tuple):
            raise TypeError(f"Unable to add object of type {type(other).__name__} to a Location.")
        dz = other[2] if len(other) > 2 else 0
        return Location(self.x + other[0], self.y + other[1], self.z + dz)

    def __mul__(self, k):
        if not isinstance(k, int):
            raise NotImplementedError
        return Location(self.x * k, self.y * k, self.z * k)

    def __eq__(self, other):
        if isinstance(other, Vector):
            other = other.compute()
        if not isinstance(other, tuple):
            raise TypeError(f"Unable to compare object of type {type(other).__name__} with a Location.")
        if len(other) == 2:
            return self.dims == 2 and self.x == other[0] and self.y == other[1]
        return self.x == other[0] and self.y == other[1] and self.z == other[2]

    __hash__ = tuple.__hash__

    def __len__(self):
        return self.dims

    def adjacent(self, world_dims):
        """The (up to four) in-bounds neighbours, in the reference's order."""
        out = []
        for step in ((1, 0), (0, 1), (-1, 0), (0, -1)):
            loc = self + Vector(*step)
            if 0 <= loc.x < world_dims[0] and 0 <= loc.y < world_dims[1]:
                out.append(loc)
        return out


class Vector:
    """Steps forward/right/backward/left (and layers) relative to a compass direction."""

    def __init__(self, forward, right, backward=0, left=0, layer=0, direction=0):
        self.forward, self.right, self.backward, self.left = forward, right, backward, left
        self.layer, self.direction = layer, direction

    def __repr__(self):
        return (f"Vector(direction={self.direction},forward={self.forward},right={self.right},"
                f"backward={self.backward},left={self.left}")

    __str__ = __repr__

    def __mul__(self, k):
        if not isinstance(k, int):
            raise NotImplementedError
        return Vector(self.forward * k, self.right * k, self.backward * k, self.left * k, self.layer * k, self.direction)

    def __add__(self, other):
        if not isinstance(other, Vector):
            raise NotImplementedError
        other.rotate(self.direction)
        return Vector(self.forward + other.forward, self.right + other.right, self.backward + other.backward,
                      self.left + other.left, self.layer + other.layer, direction=self.direction)

    def rotate(self, new_direction):
        """Re-express the same displacement relative to ``new_direction`` (in place)."""
        legs = [self.forward, self.right, self.backward, self.left]
        n = (self.direction - new_direction) % 4
        legs = legs[-n:] + legs[:-n] if n else legs
        self.forward, self.right, self.backward, self.left = legs
        self.direction = new_direction

    def compute(self) -> Location:
        f, r, b, l = _STEPS[self.direction % 4]
        a = self.forward * f[0] + self.right * r[0] + self.backward * b[0] + self.left * l[0]
        c = self.forward * f[1] + self.right * r[1] + self.backward * b[1] + self.left * l[1]
        return Location(a, c, self.layer)

    def to_tuple(self):
        """The computed offset as a plain tuple (``sorrel/location.py:317-318``)."""
        return self.compute().to_tuple()
