"""Drop-in import paths: ``import sorrel.environment`` & co. resolve to this package.

The reference is imported as ``sorrel.environment``, ``sorrel.worlds`` (re-exports ``Gridworld``,
``sorrel/worlds/__init__.py:1-2``), ``sorrel.entities`` (``Entity``, ``EmptyEntity``, ``Gem``, ``Wall``,
``sorrel/entities/__init__.py:1-2``), ``sorrel.agents`` (``Agent``, ``MovingAgent``, ``sorrel/agents/__init__.py:1``) and
the ``__init__``-less namespace packages ``sorrel.observation.observation_spec``, ``sorrel.observation.visual_field``,
``sorrel.action.action_spec``, ``sorrel.utils.helpers``, plus ``sorrel.location`` and ``sorrel.buffers`` (SURVEY.md 8 b).
``sorrel_amd`` mirrors that layout module for module, so switching is a matter of names::

    import sorrel_amd.compat
    sorrel_amd.compat.install()          # from here on ``sorrel`` IS ``sorrel_amd``
    from sorrel.environment import Environment
    from sorrel.worlds import Gridworld

or, without touching the script at all::

    python -m sorrel_amd.compat my_experiment.py --its --own --flags

``install`` registers a meta-path finder that answers every ``sorrel`` / ``sorrel.*`` import with the ``sorrel_amd``
module of the same relative name (the SAME module object: ``sorrel.environment.Environment is
sorrel_amd.environment.Environment``).  Parts of the reference outside the hot path this package rebuilds (models beyond
``BaseModel`` / ``RandomModel``, logging, visualisation, the CLI, NodeWorld, chess / iowa) do not exist here; importing
them fails with a ``ModuleNotFoundError`` that says so.  If a real ``sorrel`` distribution is importable, ``install``
refuses to shadow it unless ``force=True``.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

_TARGET = "sorrel_amd"

#: reference modules a user of the hot path imports -> all present in the mirror under the same relative name
MIRRORED = (
    "", "environment", "worlds", "worlds.gridworld", "entities", "entities.entity", "entities.basic_entities", "agents",
    "agents.agent", "observation", "observation.observation_spec", "observation.visual_field", "observation.embedding",
    "action", "action.action_spec", "utils", "utils.helpers", "location", "buffers", "models", "models.base_model",
    "examples", "examples.treasurehunt", "examples.tag", "examples.cleanup",
)

#: reference modules that are deliberately NOT rebuilt (out of scope) -> a clear error, not a stub
OUT_OF_SCOPE = {
    "utils.logging": "per-epoch scalar logging (pass any object with record_turn(epoch, loss, reward, epsilon) as `logger`)",
    "utils.visualization": "sprite rendering / GIFs",
    "cli": "the `sorrel run` launcher",
    "threadsafe": "RLock wrappers around a shared model",
    "worlds.nodeworld": "the graph world for LLM agents",
    "worlds.base_world": None,       # (the abstract World lives in sorrel_amd.worlds.gridworld; aliased below)
    "models.pytorch": "IQN / PPO / ViT policy learning",
    "models.human_player": "interactive play",
    "models.llm": "LLM clients",
    "examples.chess": "chess",
    "examples.iowa": "the Iowa gambling task",
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module
        # importlib's module_from_spec overwrites __spec__ / __loader__ / __package__ of whatever create_module returns with
        # the ALIAS's; the mirror's module must keep its own (importlib.reload and relative imports go by them)
        self._own = {k: getattr(module, k, None) for k in ("__spec__", "__loader__", "__package__", "__name__")}

    def create_module(self, spec):
        return self._module          # the mirror's own module object: classes compare identical under both names

    def exec_module(self, module):
        for k, v in self._own.items():
            if v is not None or k == "__package__":
                try:
                    setattr(module, k, v)
                except (AttributeError, TypeError):
                    pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def __init__(self, alias: str):
        self.alias = alias

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.alias and not fullname.startswith(self.alias + "."):
            return None
        rel = fullname[len(self.alias):].lstrip(".")
        if rel == "worlds.base_world":
            rel = "worlds.gridworld"
        for prefix, what in OUT_OF_SCOPE.items():
            if what and (rel == prefix or rel.startswith(prefix + ".")):
                raise ModuleNotFoundError(
                    f"{fullname} is outside the hot path sorrel_amd rebuilds ({what}); only Environment.take_turn and what "
                    "sits under it is mirrored -- see sorrel_amd.compat.MIRRORED", name=fullname)
        try:
            module = importlib.import_module(_TARGET + ("." + rel if rel else ""))
        except ModuleNotFoundError as exc:
            if exc.name and exc.name.startswith(_TARGET):
                return None          # -> "No module named 'sorrel.<x>'"
            raise
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(module), is_package=hasattr(module, "__path__"))


def installed(alias: str = "sorrel") -> bool:
    return any(isinstance(f, _AliasFinder) and f.alias == alias for f in sys.meta_path)


def install(alias: str = "sorrel", force: bool = False) -> None:
    """Make ``import <alias>...`` resolve to ``sorrel_amd...``.  Idempotent."""
    if installed(alias):
        return
    if not force:
        if alias in sys.modules and not getattr(sys.modules[alias], "__name__", "").startswith(_TARGET):
            raise ImportError(f"a different '{alias}' package is already imported in this process; pass force=True to shadow it")
        try:
            found = importlib.util.find_spec(alias)
        except (ImportError, ValueError):
            found = None
        if found is not None:
            raise ImportError(f"a real '{alias}' distribution is importable ({found.origin}); pass force=True to shadow it with sorrel_amd")
    if force:
        for name in [n for n in sys.modules if n == alias or n.startswith(alias + ".")]:
            del sys.modules[name]
    sys.meta_path.insert(0, _AliasFinder(alias))


def uninstall(alias: str = "sorrel") -> None:
    sys.meta_path[:] = [f for f in sys.meta_path if not (isinstance(f, _AliasFinder) and f.alias == alias)]
    for name in [n for n in sys.modules if n == alias or n.startswith(alias + ".")]:
        del sys.modules[name]


def main(argv=None) -> int:
    """``python -m sorrel_amd.compat script.py [args...]``: run a script written against the reference's import paths."""
    import runpy

    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
