"""Batched ``Gridworld`` behind the interface of ``sorrel/worlds/gridworld.py:10-200``.

The reference world is ``np.ndarray((H, W, L), dtype=object)`` holding ``Entity``
instances.  Here ``num_envs`` worlds live in ONE ``uint8`` tensor ``grid[E, L, H, W]``
of entity *type ids* (per-env contiguous, layer-major so a 32x32 layer is one 1 KiB
coalesced read); the per-type attributes are kept in a registry and compiled into the
engine's tables.  The methods below keep the reference's names and argument meaning;
every mutator takes an optional ``env`` (index / slice / None = all envs).

They are host-side set-up plumbing (torch indexing).  The per-turn path --
``Environment.take_turn``, ``ObservationSpec.observe``, ``MovingAgent.act`` -- runs in
the HIP kernels.
"""
from __future__ import annotations

import copy
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from sorrel_amd.entities.entity import Entity
from sorrel_amd.entities.rules import SpawnRule
from sorrel_amd.location import Location
from sorrel_amd.spec import alloc_grid, resolve_device


class World:
    """Abstract world (``sorrel/worlds/base_world.py:6-18``)."""

    def create_world(self) -> None:  # pragma: no cover - interface
        raise NotImplementedError


class TypeRegistry:
    """entity type key -> uint8 id, with one prototype entity per id."""

    def __init__(self):
        self.ids: Dict[tuple, int] = {}
        self.prototypes: List[Entity] = []
        self.version = 0

    def register(self, entity: Entity) -> int:
        key = entity.type_key()
        tid = self.ids.get(key)
        if tid is None:
            tid = len(self.prototypes)
            if tid >= 32:
                raise ValueError("more than 32 distinct entity types (engine limit SGW_MAX_TYPES)")
            self.ids[key] = tid
            proto = copy.copy(entity)
            proto._location = None
            self.prototypes.append(proto)
            self.version += 1
        return tid

    def __len__(self):
        return len(self.prototypes)


class Gridworld(World):
    """``num_envs`` independent gridworlds of ``height x width x layers`` cells.

    Attributes kept from the reference: ``height``, ``width``, ``layers``,
    ``default_entity``, ``map`` (object-array *view* of one env, read only),
    ``turn``, ``max_turns``, ``total_reward``, ``is_done``."""

    def __init__(self, height: int, width: int, layers: int, default_entity: Entity, num_envs: int = 1,
                 device=None, seed: int = 0):
        self.height, self.width, self.layers = int(height), int(width), int(layers)
        self.default_entity = default_entity
        self.num_envs = int(num_envs)
        self.device = resolve_device(device)
        self.seed = int(seed)
        self.registry = TypeRegistry()
        self.default_type = self.registry.register(default_entity)
        self.grid = alloc_grid(self.num_envs, self.layers, self.height, self.width, self.device)
        self.total_reward = torch.zeros((self.num_envs,), dtype=torch.float64, device=self.device)
        # agent bookkeeping (filled by Environment / add())
        self.agent_slots: List = []
        self.agent_pos: Optional[torch.Tensor] = None
        self.agent_layer: Optional[int] = None
        self.agent_state: Optional[torch.Tensor] = None   # [E, A] current type of each agent (interaction rules)
        self.agent_dir: Optional[torch.Tensor] = None     # [E, A] facing of each agent (Cleanup)
        # declarative reset layout (set_layout) -- None = host-built template
        self.layout = None
        self.mutations = 0          # bumped by every host-side write to the grid / positions (invalidates cached observations)
        self.create_world()
        self.turn = 0
        self.max_turns = 0
        self.is_done = False

    def mark_dirty(self) -> None:
        """Tell the Environment that ``grid`` / ``agent_pos`` were written DIRECTLY (tensor indexing instead of
        ``add`` / ``remove`` / ``move``, which call this themselves): observations rendered ahead of an agent's ``pov``
        are dropped and rendered again on demand."""
        self.mutations += 1

    # ------------------------------------------------------------------ reference API
    def create_world(self) -> None:
        """Fill every cell of every env with the default entity and zero ``total_reward``
        (``gridworld.py:47-65``)."""
        self.grid.fill_(self.default_type)
        self.total_reward.zero_()
        self.mutations += 1

    def _envsel(self, env):
        return slice(None) if env is None else env

    @staticmethod
    def _yxz(location):
        loc = tuple(location.to_tuple()) if isinstance(location, Location) else tuple(location)
        if len(loc) != 3:
            raise IndexError(f"location {loc} must be (y, x, z)")
        return int(loc[0]), int(loc[1]), int(loc[2])

    def add(self, target_location, entity: Entity, env=None) -> None:
        """Place ``entity`` at ``target_location`` (replacing what is there) in the selected envs
        (``gridworld.py:67-76``)."""
        y, x, z = self._yxz(target_location)
        entity.location = (y, x, z)
        tid = self.registry.register(entity)
        self.grid[self._envsel(env), z, y, x] = tid
        self.mutations += 1
        slot = getattr(entity, "slot", None)
        if slot is not None and self.agent_pos is not None:
            self.agent_pos[self._envsel(env), slot, 0] = y
            self.agent_pos[self._envsel(env), slot, 1] = x
            if self.agent_layer is None:
                self.agent_layer = z
            elif self.agent_layer != z:
                raise ValueError("all agents must live on one layer")

    def remove(self, target_location, env: int = 0) -> Entity:
        """Refill the cell with the default entity; returns the previous occupant's prototype
        (``gridworld.py:78-93``)."""
        y, x, z = self._yxz(target_location)
        prev = self.observe((y, x, z), env=env)
        self.grid[self._envsel(env), z, y, x] = self.default_type
        self.mutations += 1
        return prev

    def move(self, entity: Entity, new_location, env: int = 0) -> bool:
        """Host-side single-env move with the reference's semantics (``gridworld.py:95-122``).
        The per-turn moves of agents do not come through here (HIP kernel)."""
        y, x, z = self._yxz(new_location)
        if not self.observe((y, x, z), env=env).passable:
            return False
        oy, ox, oz = self._yxz(entity.location)
        self.grid[env, z, y, x] = self.registry.register(entity)
        self.grid[env, oz, oy, ox] = self.default_type
        self.mutations += 1
        entity.location = (y, x, z)
        slot = getattr(entity, "slot", None)
        if slot is not None and self.agent_pos is not None:
            self.agent_pos[env, slot, 0], self.agent_pos[env, slot, 1] = y, x
        return True

    def observe(self, target_location, env: int = 0) -> Entity:
        """The entity (type prototype, with ``location`` set) at a location of one env (``gridworld.py:124-133``)."""
        y, x, z = self._yxz(target_location)
        if y >= self.height or x >= self.width or z >= self.layers:
            raise IndexError(f"index {(y, x, z)} is out of bounds")   # numpy raises at the high edge
        tid = int(self.grid[env, z, y, x])                              # negative indices wrap, as in numpy
        e = copy.copy(self.registry.prototypes[tid])
        e.location = (y % self.height, x % self.width, z % self.layers)
        return e

    def observe_all_layers(self, target_location, env: int = 0) -> List[Entity]:
        loc = tuple(target_location)
        return [self.observe((loc[0], loc[1], i), env=env) for i in range(self.layers)]

    def valid_location(self, index) -> bool:
        """``gridworld.py:153-179``: rank mismatch raises IndexError, negatives are invalid."""
        if isinstance(index, Location):
            index = index.to_tuple()
        shape = (self.height, self.width, self.layers)
        if len(index) != len(shape):
            raise IndexError(f"Index {index} and world shape {shape} must be the same length.")
        return min(index) >= 0 and all(i < s for s, i in zip(shape, index))

    def get_entities_of_kind(self, kind: str, env: int = 0) -> List[Entity]:
        """All entities of one env whose ``kind`` matches, in ``ndenumerate`` order (``gridworld.py:181-196``)."""
        out = []
        g = self.grid[env].cpu().numpy()
        for (y, x, z), _ in np.ndenumerate(np.empty((self.height, self.width, self.layers), dtype=np.uint8)):
            proto = self.registry.prototypes[int(g[z, y, x])]
            if proto.kind == kind:
                e = copy.copy(proto)
                e.location = (y, x, z)
                out.append(e)
        return out

    @property
    def map(self) -> np.ndarray:
        """Object array ``(H, W, L)`` of entity prototypes for env 0 -- a read-only snapshot for
        code that inspects ``world.map`` (the reference's storage, ``gridworld.py:56``)."""
        return self.env_map(0)

    def env_map(self, env: int) -> np.ndarray:
        g = self.grid[env].cpu().numpy()
        out = np.empty((self.height, self.width, self.layers), dtype=object)
        for (y, x, z), _ in np.ndenumerate(out):
            e = copy.copy(self.registry.prototypes[int(g[z, y, x])])
            e.location = (y, x, z)
            out[y, x, z] = e
        return out

    # ------------------------------------------------------------------ batched additions
    def set_layout(self, layer_fill: Sequence[Entity], layer_border: Sequence[Optional[Entity]],
                   dense_prob: float = 0.0, dense_choices: Sequence[Entity] = ()) -> None:
        """Declare ``populate_environment`` for the on-device reset kernel: per layer a fill
        entity and an optional border entity (walls), optional Bernoulli pre-seeding of the
        agent layer's interior, agents on distinct random interior cells of their layer
        (the Treasurehunt populate, ``sorrel/examples/treasurehunt/env.py:114-147``)."""
        if len(layer_fill) != self.layers or len(layer_border) != self.layers:
            raise ValueError("set_layout needs one fill and one border entry per layer")
        self.layout = dict(
            fill=[self.registry.register(e) for e in layer_fill],
            border=[255 if e is None else self.registry.register(e) for e in layer_border],
            dense_prob=float(dense_prob),
            dense=[self.registry.register(e) for e in dense_choices],
        )

    def set_template(self, entities: np.ndarray) -> None:
        """Copy one host-built map -- an object array ``(H, W, L)`` of entities, the reference's
        ``world.map`` layout -- into every env with a single upload (instead of one ``add`` per cell)."""
        if entities.shape != (self.height, self.width, self.layers):
            raise IndexError(f"template shape {entities.shape} != world shape {(self.height, self.width, self.layers)}")
        ids = np.empty((self.layers, self.height, self.width), dtype=np.uint8)
        for (y, x, z), e in np.ndenumerate(entities):
            ids[z, y, x] = self.default_type if e is None else self.registry.register(e)
        self.grid.copy_(torch.from_numpy(ids).to(self.device).expand_as(self.grid))
        self.mutations += 1

    def scatter_random(self, cells: Sequence, count: int, entity: Entity, generator: torch.Generator) -> torch.Tensor:
        """In every env put ``entity`` on ``count`` of the candidate ``cells`` [(y, x, z)], drawn without
        replacement per env (the batched ``np.random.choice(len(points), size=count, replace=False)``
        of the reference's populate code).  Returns the chosen cell indices ``[E, count]``."""
        if count > len(cells):
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        pts = torch.tensor([tuple(int(v) for v in c) for c in cells], dtype=torch.long).reshape(-1, 3)
        keys = torch.rand((self.num_envs, len(cells)), generator=generator)
        pick = keys.argsort(dim=1)[:, :count]                                   # [E, count]
        tid = self.registry.register(entity)
        e_idx = torch.arange(self.num_envs)[:, None].expand_as(pick)
        sel = pts[pick]                                                          # [E, count, 3]
        dev = self.device
        self.grid[e_idx.to(dev), sel[..., 2].to(dev), sel[..., 0].to(dev), sel[..., 1].to(dev)] = tid
        self.mutations += 1
        return pick

    def spawn_rule_of(self, proto: Entity):
        """(prob, [type ids]) of a prototype's SpawnRule, resolving callables against this world."""
        rule = proto.transition_rule
        if not isinstance(rule, SpawnRule):
            raise ValueError(f"{type(proto).__name__}: unsupported transition rule {rule!r}")
        prob, choices = rule.resolve(self)
        return prob, [self.registry.register(c) for c in choices]
