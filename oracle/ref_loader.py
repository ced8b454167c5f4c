"""TEST INFRASTRUCTURE ONLY -- loader that imports the *reference* Sorrel package.

This file is inert without ``/root/reference`` (it exists only in the build
container, never on the GPU box).  It is used by ``oracle/make_golden.py`` to
generate the golden fixtures under ``tests/golden/`` and by the ``not gpu``
tests that cross-check the CPU restatement against the running reference.

The reference needs Python >= 3.12 syntax (PEP 695 class generics,
``sorrel/entities/entity.py:9``; PEP 646 star-subscripts,
``sorrel/observation/visual_field.py:49``) while this image ships 3.10, and it
imports ``omegaconf`` / ``IPython`` / ``tensorboard`` which are absent.  The
loader below reads the reference's source *in memory*, applies four mechanical,
semantics-free down-levelling rewrites at import time and installs stub modules
for the three missing third-party packages.  Nothing is written to disk and no
reference source is copied into this repository.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import importlib.util
import os
import re
import sys
import types

REFERENCE_ROOT = os.environ.get("SORREL_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sorrel"))


# --------------------------------------------------------------------------- #
# source down-levelling (3.12 -> 3.10), applied to text in memory only
# --------------------------------------------------------------------------- #
_CLASS_LINE = re.compile(r"^(\s*class\s+.*)$")


def _strip_brackets(line: str) -> str:
    """Remove every balanced ``[...]`` group from a ``class`` header line."""
    out, depth = [], 0
    for ch in line:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        elif depth == 0:
            out.append(ch)
    return "".join(out)


def downlevel(text: str, path: str) -> str:
    lines = text.split("\n")
    for i, line in enumerate(lines):
        if _CLASS_LINE.match(line) and "[" in line:
            lines[i] = _strip_brackets(line)
        if "new[:, *index]" in line:
            lines[i] = line.replace("new[:, *index]", "new[(slice(None), *index)]")
    text = "\n".join(lines)
    # nested same-quote f-strings (3.12 only) in two non-hot-path files
    if path.endswith("utils/logging.py") or path.endswith("iowa/main.py"):
        text = re.sub(
            r'f"([^"\n]*)\{([^{}\n]*)"([^"\n]*)"([^{}\n]*)\}([^"\n]*)"',
            lambda m: 'f"%s{%s\'%s\'%s}%s"' % m.groups(),
            text,
        )
    # annotations mention the removed type params: make them lazy, keeping
    # line numbers intact.
    if "from __future__ import annotations" not in text:
        text = _add_future(text)
    return text


_DOCSTRING = re.compile(r'\A((?:[ \t]*(?:#[^\n]*)?\n)*)[ \t]*[rRuU]?("""|\'\'\')')


def _add_future(text: str) -> str:
    """Insert the __future__ import without shifting any line number."""
    m = _DOCSTRING.match(text)
    if m:  # module docstring first: append to the line that closes it
        q = m.group(2)
        end = text.find(q, m.end())
        if end >= 0:
            end += 3
            return text[:end] + ";from __future__ import annotations" + text[end:]
    m = re.match(r"\A((?:[ \t]*(?:#[^\n]*)?\n)*)", text)
    head = m.group(1) if m else ""
    return head + "from __future__ import annotations;" + text[len(head):]


class _Loader(importlib.abc.SourceLoader):
    def __init__(self, fullname: str, path: str):
        self.fullname, self.path = fullname, path

    def get_filename(self, fullname):
        return self.path

    def get_data(self, path):
        with open(path, "rb") as fh:
            data = fh.read()
        if path.endswith(".py"):
            return downlevel(data.decode("utf-8"), path).encode("utf-8")
        return data

    def source_to_code(self, data, path, *, _optimize=-1):
        return compile(data, path, "exec", dont_inherit=True, optimize=_optimize)

    # never touch bytecode caches
    def path_stats(self, path):
        raise OSError

    def set_data(self, path, data):
        return None


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "sorrel" and not fullname.startswith("sorrel."):
            return None
        rel = fullname.replace(".", os.sep)
        base = os.path.join(REFERENCE_ROOT, rel)
        if os.path.isdir(base):
            init = os.path.join(base, "__init__.py")
            if os.path.isfile(init):
                return importlib.util.spec_from_file_location(
                    fullname, init, loader=_Loader(fullname, init),
                    submodule_search_locations=[base],
                )
            spec = importlib.machinery.ModuleSpec(fullname, None, is_package=True)
            spec.submodule_search_locations = [base]
            return spec
        if os.path.isfile(base + ".py"):
            return importlib.util.spec_from_file_location(
                fullname, base + ".py", loader=_Loader(fullname, base + ".py")
            )
        return None


# --------------------------------------------------------------------------- #
# stubs for third-party packages the reference imports but this image lacks
# --------------------------------------------------------------------------- #
class DictConfig(dict):
    """Attribute-style dict: the subset of omegaconf.DictConfig the hot path uses."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = DictConfig(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _install_stubs() -> None:
    if "omegaconf" not in sys.modules:
        m = types.ModuleType("omegaconf")

        class OmegaConf:
            @staticmethod
            def create(d=None):
                return DictConfig(d or {})

            @staticmethod
            def from_dotlist(items):
                root: dict = {}
                for it in items:
                    k, v = it.split("=", 1)
                    cur = root
                    parts = k.split(".")
                    for p in parts[:-1]:
                        cur = cur.setdefault(p, {})
                    cur[parts[-1]] = v
                return DictConfig(root)

        m.DictConfig, m.OmegaConf = DictConfig, OmegaConf
        sys.modules["omegaconf"] = m
    if "IPython" not in sys.modules:
        ip = types.ModuleType("IPython")
        disp = types.ModuleType("IPython.display")
        disp.clear_output = lambda *a, **k: None
        ip.display = disp
        sys.modules["IPython"], sys.modules["IPython.display"] = ip, disp
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        tb = types.ModuleType("torch.utils.tensorboard")
        wr = types.ModuleType("torch.utils.tensorboard.writer")

        class SummaryWriter:  # pragma: no cover - never used on the hot path
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

        wr.SummaryWriter = tb.SummaryWriter = SummaryWriter
        tb.writer = wr
        sys.modules["torch.utils.tensorboard"] = tb
        sys.modules["torch.utils.tensorboard.writer"] = wr


_installed = False


def install() -> None:
    """Make ``import sorrel`` resolve to the reference (idempotent)."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.meta_path.insert(0, _Finder())
    _installed = True
