"""The loops that can play an ``Environment.take_turn()`` (round 6: moved out of ``environment.py``, behind the same ``Environment`` methods) -- one module
each, mixed into ``sorrel_amd.environment.Environment``:

* ``speculative``  many policy-driven agents: one batched policy evaluation per pass (``sgw_turn_resolve``; any agent rule: ``sgw_verify_rows``)
* ``mixed``        agents that hold different observation / action specs (one engine handle per distinct pair)
* ``policy``       the eager policy-driven turn: windows once (+ ``sgw_act`` per agent), the fast loop of agents with the standard hooks
* ``recorded``     that turn as ONE graph replay (``capture_turn``, ``sgw_turn_*``)

``Environment.turn_plan()`` says which one plays the next turn and why the faster ones do not apply.  The fused one-launch turn (device-drawn or
given actions) and the reference's ``Agent.transition`` loop live in ``environment.py`` itself.  Reference: ``sorrel/environment.py:81-93``,
``sorrel/agents/agent.py:155-173``."""
from sorrel_amd.turns.mixed import MixedSpecTurns
from sorrel_amd.turns.policy import PolicyTurns, _FastPolicyTurn
from sorrel_amd.turns.recorded import CapturedTurn, RecordedTurns
from sorrel_amd.turns.speculative import SpeculativeTurns

__all__ = ["SpeculativeTurns", "MixedSpecTurns", "PolicyTurns", "RecordedTurns", "CapturedTurn", "_FastPolicyTurn"]
