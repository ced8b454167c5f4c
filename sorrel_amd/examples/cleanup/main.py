"""Run batched Cleanup rollouts: ``python -m sorrel_amd.examples.cleanup.main`` (the counterpart of
``sorrel/examples/cleanup/main.py`` with the defaults of ``configs/config.yaml``; random agents)."""
from sorrel_amd.examples.cleanup.entities import EmptyEntity
from sorrel_amd.examples.cleanup.env import CleanupEnv
from sorrel_amd.examples.cleanup.world import CleanupWorld


def make_config(height=21, width=31, num_agents=10, vision=5, beam_radius=3, epochs=2, max_turns=100):
    return {
        "experiment": {"epochs": epochs, "max_turns": max_turns, "record_period": 50},
        "env": {"height": height, "width": width, "layers": 3, "pollution_threshold": 0.5, "initial_apples": 20,
                "apple_spawn_chance": 0.002, "pollution_spawn_chance": 0.009, "mode": "DEFAULT"},
        "agent": {"agent": {"num": num_agents, "beam_radius": beam_radius, "obs": {"vision": vision, "embeddings": 3}}},
    }


if __name__ == "__main__":
    config = make_config()
    world = CleanupWorld(config=config, default_entity=EmptyEntity(), num_envs=4096)
    env = CleanupEnv(world, config)
    for epoch, m in enumerate(env.run_experiment()):
        print(f"epoch {epoch}: mean total_reward over {int(m['envs'])} envs = {m['mean_total_reward']:.3f}, "
              f"polluted fraction of the river = {float(world.measure_pollution().mean()):.3f}")
