"""Run batched Treasurehunt rollouts: ``python -m sorrel_amd.examples.treasurehunt.main``."""
from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld


def make_config(height=21, width=21, num_agents=2, radius=2, spawn_prob=0.005, epochs=2, max_turns=100):
    return {
        "experiment": {"epochs": epochs, "max_turns": max_turns, "record_period": 50},
        "model": {"agent_vision_radius": radius, "num_agents": num_agents},
        "world": {"height": height, "width": width, "gem_value": 10, "food_value": 5, "bone_value": -10,
                  "spawn_prob": spawn_prob},
    }


if __name__ == "__main__":
    config = make_config()
    world = TreasurehuntWorld(config=config, default_entity=EmptyEntity(), num_envs=4096)
    env = TreasurehuntEnv(world, config)
    for epoch, m in enumerate(env.run_experiment()):
        print(f"epoch {epoch}: mean total_reward over {int(m['envs'])} envs = {m['mean_total_reward']:.3f}")
