"""Observation specs with the interface of ``sorrel/observation/observation_spec.py``.

``OneHotObservationSpec.observe`` is the reference's ``visual_field`` hot spot
(81-83 % of a reference ``take_turn``): here it is a gather from LDS in the HIP
kernels (``sgw_observe`` / fused into ``sgw_step``); this class only holds the
appearance table (``entity_map``) the engine is compiled from.
"""
from __future__ import annotations

import colorsys
from abc import abstractmethod
from typing import Dict, Optional, Sequence

import numpy as np

from sorrel_amd.utils.helpers import one_hot_encode


class ObservationSpec:
    """``entity_map`` (kind -> appearance), ``vision_radius``, ``full_view``, ``input_size``,
    ``fill_entity_kind`` -- same constructor and errors as ``observation_spec.py:30-59``."""

    entity_map: Dict[str, np.ndarray]
    vision_radius: int
    full_view: bool
    input_size: Sequence[int]
    fill_entity_kind: str

    def __init__(self, entity_list, full_view: bool, vision_radius: Optional[int] = None,
                 env_dims: Optional[Sequence[int]] = None, fill_entity_kind: str = "Wall"):
        if full_view and not isinstance(env_dims, Sequence):
            raise TypeError("env_dims must be provided when full_view is true.")
        elif not full_view and not isinstance(vision_radius, int):
            raise TypeError("vision_radius must be provided when full_view is false.")
        self.full_view = full_view
        self.vision_radius = vision_radius if vision_radius else 0
        self.entity_list = list(entity_list)
        self.entity_map = self.generate_map(self.entity_list)
        self.input_size = (1,)
        self.fill_entity_kind = fill_entity_kind

    @abstractmethod
    def generate_map(self, entity_list):
        ...

    @abstractmethod
    def observe(self, world, location=None):
        ...

    def override_entity_map(self, entity_map) -> None:
        """Custom appearances (e.g. several classes sharing one); non one-hot tables take the
        engine's general float64-layer-sum path."""
        self.entity_map = entity_map

    def override_input_size(self, input_size) -> None:
        self.input_size = input_size

    def invalidate(self) -> None:
        """Call after editing an appearance vector of ``entity_map`` IN PLACE: the engine's tables are compiled from the
        map once per (map object, radius, fill kind) and cached on the spec; replacing the map (``override_entity_map``)
        or changing ``vision_radius`` / ``fill_entity_kind`` is seen by itself, an in-place edit is not."""
        self.__dict__.pop("_sgw_key", None)

    @property
    def num_channels(self) -> int:
        return len(next(iter(self.entity_map.values())))

    #: post-processing of the layer sum the engine applies (0 = none); see include/sgw.h SGW_OBS_POST_*
    obs_post = 0

    def _engine_observe(self, world, location):
        if not self.full_view and location is None:
            raise TypeError(
                "location not provided when full_view is false. Please provide the location of the observer.")
        env = getattr(world, "_environment", None)
        if env is None:
            raise RuntimeError("the world is not attached to an Environment (the engine is compiled there)")
        if self.full_view:
            return env._full_view(self, location)
        return env._observe(location, self)


class OneHotObservationSpec(ObservationSpec):
    """One-hot egocentric observations (``observation_spec.py:116-205``)."""

    def __init__(self, entity_list, full_view: bool, vision_radius: Optional[int] = None,
                 env_dims: Optional[Sequence[int]] = None, fill_entity_kind: str = "Wall"):
        super().__init__(entity_list, full_view, vision_radius, env_dims, fill_entity_kind)
        if self.full_view:
            self.input_size = (len(entity_list), *env_dims)
        else:
            v = 2 * self.vision_radius + 1
            self.input_size = (len(entity_list), v, v)

    def generate_map(self, entity_list):
        """kind i of ``entity_list`` -> one-hot(i); ``"EmptyEntity"`` -> all zeros but it still
        occupies its channel index (``observation_spec.py:152-173``)."""
        n = len(entity_list)
        return {k: (np.zeros(n) if k == "EmptyEntity" else one_hot_encode(i, n)) for i, k in enumerate(entity_list)}

    def observe(self, world, location=None):
        """Egocentric observation.

        ``location`` may be an agent (or its slot index): returns the batched float32 tensor
        ``[E, C, V, V]`` of that agent in every env (K1, ``sgw_observe``).  A plain
        ``(y, x, z)`` tuple observes from that cell in every env.  With ``full_view`` the
        whole map is returned ``[E, C, H, W]`` (appearance summed over layers,
        ``visual_field.py:41-55``)."""
        return self._engine_observe(world, location)


class RGBObservationSpec(ObservationSpec):
    """RGB image observations (``observation_spec.py:386-483``): the same window gather with an
    RGB appearance per kind (summed over layers), then ``np.clip(obs, 0, 255) / 255`` -- done by
    the kernels' float64 path (``SGW_OBS_POST_CLIP255_DIV255``)."""

    obs_post = 1

    def __init__(self, entity_list, full_view: bool, vision_radius: Optional[int] = None,
                 env_dims: Optional[Sequence[int]] = None, fill_entity_kind: str = "Wall"):
        super().__init__(entity_list, full_view, vision_radius, env_dims, fill_entity_kind)
        if self.full_view:
            self.input_size = (3, *env_dims)
        else:
            v = 2 * self.vision_radius + 1
            self.input_size = (3, v, v)

    def generate_map(self, entity_list):
        """``"EmptyEntity"`` -> black; the other kinds get evenly spaced hues at full saturation
        and value, as uint8 triples (``observation_spec.py:425-450``)."""
        others = [e for e in entity_list if e != "EmptyEntity"]
        out, k = {}, 0
        for kind in entity_list:
            if kind == "EmptyEntity":
                out[kind] = np.zeros(3, dtype=np.uint8)
            else:
                rgb = colorsys.hsv_to_rgb(k / max(1, len(others)), 1.0, 1.0)
                out[kind] = np.array([int(c * 255) for c in rgb], dtype=np.uint8)
                k += 1
        return out

    def observe(self, world, location=None):
        return self._engine_observe(world, location)
