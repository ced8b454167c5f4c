"""Run batched Tag rollouts: ``python -m sorrel_amd.examples.tag.main`` (``sorrel/examples/tag/main.py`` defaults:
11 x 11 single-layer world, 5 agents, vision 4, 20 turns; random agents)."""
from sorrel_amd.entities import EmptyEntity
from sorrel_amd.examples.tag.env import TagEnv
from sorrel_amd.worlds import Gridworld

if __name__ == "__main__":
    config = {
        "experiment": {"epochs": 2, "max_turns": 20, "record_period": 50},
        "agent": {"num_agents": 5, "vision_radius": 4},
        "world": {"height": 11, "width": 11, "layers": 1},
    }
    world = Gridworld(**config["world"], default_entity=EmptyEntity(), num_envs=4096)
    env = TagEnv(world, config)
    for epoch, m in enumerate(env.run_experiment()):
        print(f"epoch {epoch}: mean total_reward over {int(m['envs'])} envs = {m['mean_total_reward']:.3f}")
