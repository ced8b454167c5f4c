"""The epoch loops around ``take_turn``: ``Environment.run_experiment`` / ``generate_memories`` (``sorrel/environment.py:108-300``) and the
world-state checkpoint the reference leaves as a TODO (``sorrel/environment.py:107``)."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from sorrel_amd.agents.agent import Agent


class EpochLoops:
    """Mixed into ``sorrel_amd.environment.Environment``."""

    def _stock_take_turn(self) -> bool:
        """Nobody has overridden ``take_turn`` (the one-call and recorded forms of the epoch loops stand in for it only then)."""
        from sorrel_amd.environment import Environment

        return type(self).take_turn is Environment.take_turn

    # ------------------------------------------------------------------ model hooks (overridable, environment.py:95-105)
    def _model_start_epoch_action(self, agent: Agent, epoch: int):
        agent.model.start_epoch_action(epoch=epoch)

    def _model_end_epoch_action(self, agent: Agent, epoch: int):
        agent.model.end_epoch_action(epoch=epoch)

    def _model_train_step(self, agent: Agent):
        return agent.model.train_step()

    # ------------------------------------------------------------------ epoch loops (environment.py:108-300)
    def _output_dir(self, output_dir) -> Path:
        if output_dir is None:
            exp = self.config.experiment
            output_dir = Path(exp.output_dir) if hasattr(exp, "output_dir") or "output_dir" in exp else Path("./data/")
        output_dir = Path(output_dir)
        os.makedirs(output_dir, exist_ok=True)
        return output_dir

    def _cfg_model(self, key, default=None):
        model = getattr(self.config, "model", None) if not isinstance(self.config, dict) else self.config.get("model")
        if model is None:
            return default
        try:
            return model[key] if key in model else default
        except TypeError:
            return getattr(model, key, default)

    def run_experiment(self, animate: bool = False, logging: bool = True, logger=None, output_dir=None,
                       epochs: Optional[int] = None, max_turns: Optional[int] = None, all_reduce: bool = True):
        """``for epoch in range(epochs + 1)``: reset -> start-of-epoch hooks -> ``max_turns`` x take_turn ->
        ``world.is_done = True`` -> end-of-epoch hooks -> ``train_step`` per agent (the loss logged is the LAST
        agent's, as in the reference: assignment, not a sum) -> ``logger.record_turn(epoch, loss, reward, epsilon)``
        -> epsilon decay -> model checkpoint every ``record_period`` epochs when ``config.model.save_weights``
        (``sorrel/environment.py:148-211``).  The reward logged is the mean of ``world.total_reward`` over ALL
        envs of ALL ranks (the one RCCL all-reduce); the per-epoch metric dicts are returned.  ``animate`` is
        accepted for signature compatibility; sprite rendering is outside this engine."""
        from sorrel_amd import distributed as D

        exp = self.config.experiment
        epochs = int(exp.epochs) if epochs is None else epochs
        max_turns = int(exp.max_turns) if max_turns is None else max_turns
        record_period = int(exp.record_period) if (hasattr(exp, "record_period") or "record_period" in exp) else 1
        save_weights = bool(self._cfg_model("save_weights", False))
        decay = self._cfg_model("epsilon_decay", None)
        out_dir = self._output_dir(output_dir) if save_weights else None
        capture = bool(self.capture_turns or (exp.get("capture_turns", False) if hasattr(exp, "get") else getattr(exp, "capture_turns", False)))
        history = []
        for epoch in range(epochs + 1):
            self.reset()
            for agent in self.agents:
                self._model_start_epoch_action(agent, epoch)
            if all(getattr(a.model, "device_random", False) for a in self.agents) and not self.stop_if_done \
                    and self._stock_take_turn():
                self.rollout(max_turns - self.turn)        # the whole epoch in one engine call (a subclass that overrides
                                                           # take_turn gets its per-turn loop below, as in the reference)
            elif capture and self._captured is None and not self.stop_if_done and self._stock_take_turn() \
                    and max_turns - self.turn > 2:
                capture = self.capture_turn(warmup=2) is not None      # (two real turns of this epoch; not tried again when it fails)
            while self.turn < max_turns:
                self.take_turn()
                if self.world.is_done and self.stop_if_done:
                    break
            self.world.is_done = True
            self.raise_on_status()
            m = D.rollout_metrics(self._ensure_engine(), all_reduce=all_reduce)
            for agent in self.agents:
                self._model_end_epoch_action(agent, epoch)
            total_loss = 0
            for agent in self.agents:
                total_loss = self._model_train_step(agent)
            m["loss"] = float(total_loss) if total_loss is not None else 0.0
            m["epsilon"] = float(getattr(self.agents[0].model, "epsilon", 0.0))
            history.append(m)
            if logging and logger is not None:
                logger.record_turn(epoch, total_loss, m["mean_total_reward"], m["epsilon"])
            for i, agent in enumerate(self.agents):
                if decay is not None:
                    agent.model.epsilon_decay(float(decay))
                if epoch % record_period == 0 and save_weights and hasattr(agent.model, "save"):
                    os.makedirs(out_dir / "checkpoints", exist_ok=True)
                    agent.model.save(out_dir / "checkpoints" / f"epoch{epoch}-agent-{i}.pkl")
        return history

    def generate_memories(self, num_games: int = 1000, animate: bool = False, output_dir=None,
                          record_positions: bool = False):
        """Play ``num_games`` games of ``max_turns`` turns with the existing models and write one replay file per
        agent, ``<output_dir>/memories/agent{i}.npz`` (``sorrel/environment.py:213-300``).

        File format = the reference's ``SavedGames.save`` (``sorrel/buffers.py:361-379``): ``states`` float32
        ``[N, *obs_shape]``, ``actions`` int64 ``[N]``, ``rewards`` / ``dones`` float32 ``[N]``, ``positions`` int64
        ``[N, 2]``, ``n_frames``, ``idx`` -- the reference's ``Buffer.load`` reads it.  The batch is laid out env
        by env: rows ``[e * G * T, (e + 1) * G * T)`` are env ``e``'s ``G`` games of ``T`` turns in play order, i.e.
        what the reference would have saved for that one world.

        Policy-driven agents (phased turns): after every game the agent's whole ``model.memory`` is appended with
        ``add_from_buffer``, the reference's own call -- so, as there, ``positions`` are stored only if that memory
        carries them, and a memory the model does not clear per game is appended again from its start
        (``sorrel/environment.py:297``, ``buffers.py:71-99``).  Device-random models (``RandomModel``): the turns run
        fused, the step kernel writes the observations straight into a device ring (``collect``) and every game is
        appended once; ``positions`` stay zero unless ``record_positions`` (then: each agent's cell after its move)."""
        from sorrel_amd.buffers import SavedGames, TurnBuffer

        out_dir = self._output_dir(output_dir)
        T = int(self.config.experiment.max_turns)
        E, A = self.num_envs, len(self.agents)
        saved = []
        for agent in self.agents:
            n_frames = getattr(agent.model, "n_frames", 1)
            obs_shape = tuple(agent.observation_spec.input_size)
            saved.append(SavedGames(capacity=num_games * T, obs_shape=obs_shape, n_frames=n_frames, num_envs=E,
                                    device="cpu", positions=(2,)))
            if hasattr(agent.model, "eval"):
                agent.model.eval()
        device_random = all(getattr(a.model, "device_random", False) for a in self.agents)
        # one engine call per game -- unless a subclass overrides take_turn: the reference's loop goes through take_turn
        # every turn (sorrel/environment.py:266-282), so an override (per-turn logging, extra world logic) must be called
        one_call = device_random and self._stock_take_turn()
        exp = self.config.experiment
        capture = bool(self.capture_turns or (exp.get("capture_turns", False) if hasattr(exp, "get") else getattr(exp, "capture_turns", False)))
        ring = None
        for game in range(num_games):
            self.reset()
            for agent in self.agents:
                self._model_start_epoch_action(agent, game)
            eng = self._ensure_engine()
            if device_random:
                if ring is None:
                    ring = TurnBuffer(T, E, eng.spec.obs_shape, device=eng.device, obs_dtype=eng.obs_dtype,
                                      positions=record_positions)
                ring.clear()
                if one_call:
                    self.collect(T, ring)
                else:
                    while self.turn < T:
                        self.take_turn()
                        eng = self._ensure_engine()
                        ring.obs[ring.slot()].copy_(eng.obs)
                        ring.commit(eng.actions, eng.rewards, eng.agent_pos)
                        if self.world.is_done and self.stop_if_done:
                            break
                n = len(ring)
                for a, sg in enumerate(saved):
                    st, ac, rw, dn = ring.agent_view(a)
                    sg.add_turns(st[:n], ac[:n], rw[:n], dn[:n], positions=None if ring.positions is None else ring.positions[:n, :, a])
            else:
                if capture and self._captured is None and not self.stop_if_done and self._stock_take_turn() and T - self.turn > 2:
                    capture = self.capture_turn(warmup=2) is not None      # (capture_turns: as in run_experiment)
                while self.turn < T:
                    self.take_turn()
                    if self.world.is_done and self.stop_if_done:
                        break
            self.world.is_done = True
            self.raise_on_status()
            for agent, sg in zip(self.agents, saved):
                self._model_end_epoch_action(agent, game)
                if not device_random:
                    sg.add_from_buffer(agent.model.memory)
        os.makedirs(out_dir / "memories", exist_ok=True)
        paths = []
        for i, sg in enumerate(saved):
            paths.append(out_dir / "memories" / f"agent{i}.npz")
            sg.save(paths[-1])
        return paths

    # ------------------------------------------------------------------ world-state checkpoint (the reference leaves
    # "# TODO: ability to save/load?" at sorrel/environment.py:107)
    def state_dict(self) -> dict:
        """Everything a rollout needs to continue bit-exactly: the grid, agent positions, ``total_reward``, the
        per-agent state / facing tensors, the epoch / turn counters, the RNG seed and the first global env id."""
        w = self.world
        eng = self._ensure_engine()
        sd = dict(version=1, grid=w.grid.cpu().clone(), agent_pos=w.agent_pos.cpu().clone(),
                  total_reward=w.total_reward.cpu().clone(), epoch=int(self.epoch), turn=int(self.turn),
                  seed=int(w.seed), first_env_id=int(getattr(w, "first_env_id", 0)), num_envs=int(w.num_envs),
                  shape=(w.layers, w.height, w.width), is_done=bool(w.is_done),
                  type_names=[type(p).__name__ for p in w.registry.prototypes])
        if eng.agent_state is not None:
            sd["agent_state"] = eng.agent_state.cpu().clone()
        if eng.agent_dir is not None:
            sd["agent_dir"] = eng.agent_dir.cpu().clone()
        return sd

    def load_state_dict(self, sd: dict) -> None:
        w = self.world
        eng = self._ensure_engine()
        if tuple(sd["shape"]) != (w.layers, w.height, w.width) or int(sd["num_envs"]) != w.num_envs:
            raise ValueError("checkpoint was taken from a world of another shape or batch size")
        if int(sd["seed"]) != int(w.seed) or int(sd["first_env_id"]) != int(getattr(w, "first_env_id", 0)):
            raise ValueError("checkpoint was taken with another seed / first global env id: the rollout would not continue bit-exactly")
        if list(sd["type_names"]) != [type(p).__name__ for p in w.registry.prototypes]:
            raise ValueError("checkpoint was taken with another entity type table")
        w.grid.copy_(sd["grid"].to(w.device))
        w.agent_pos.copy_(sd["agent_pos"].to(w.device))
        w.total_reward.copy_(sd["total_reward"].to(w.device))
        if "agent_state" in sd and eng.agent_state is not None:
            eng.agent_state.copy_(sd["agent_state"].to(w.device))
        if "agent_dir" in sd and eng.agent_dir is not None:
            eng.agent_dir.copy_(sd["agent_dir"].to(w.device))
        self.epoch, self.turn = int(sd["epoch"]), int(sd["turn"])
        eng.epoch, eng.turn = self.epoch, self.turn
        w.is_done = bool(sd.get("is_done", False))
        w.mutations += 1
        self._fresh_obs = None

    def save_checkpoint(self, path) -> None:
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path) -> None:
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))   # tensors and plain values only
