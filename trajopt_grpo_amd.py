"""Import alias: the package directory is `trajopt-grpo_amd/` (hyphen), which `import` cannot spell.

`import trajopt_grpo_amd` loads that directory as the package `trajopt_grpo_amd`
(submodules included: `trajopt_grpo_amd.rollout`, ...)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trajopt-grpo_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
