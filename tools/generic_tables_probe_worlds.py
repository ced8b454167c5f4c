"""Synthetic worlds with their own entity sets (shared by the probes)."""
import numpy as np
from sorrel_amd.spec import WorldSpec, action_deltas


def move_world(h, w, layers, channels, a, r, seed=3):
    T = max(6, min(channels + 1, 12))
    app = np.zeros((T, channels))
    for t in range(1, T):
        app[t, (t * 5 + 1) % channels] = 1.0
    dy, dx = action_deltas(["up", "down", "left", "right", "stay"])
    return WorldSpec(height=h, width=w, layers=layers, num_agents=a, vision_radius=r, num_channels=channels, agent_layer=layers - 1,
                     default_type=0, fill_type=1, action_dy=dy, action_dx=dx, agent_type=[T - 1] * a,
                     type_value=[0.0, -1.0, 10.0, 5.0, -10.0] + [1.0] * (T - 6) + [0.0],
                     type_passable=[1, 0, 1, 1, 1] + [1] * (T - 6) + [0], type_rule=[1] + [0] * (T - 1),
                     spawn_prob=[0.005] + [0.0] * (T - 1), spawn_choices=[[2, 3, 4]] + [[] for _ in range(T - 1)],
                     appearance=app, seed=seed, layer_fill_type=[0] * layers, layer_border_type=[1] * layers)
