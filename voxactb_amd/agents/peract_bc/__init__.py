"""`agents.peract_bc` as run_seed_fn.py uses it: `peract_bc.launch_utils.create_replay / fill_multi_task_replay /
create_agent` (reference peract/agents/peract_bc/__init__.py:1, run_seed_fn.py:107-129)."""
from . import launch_utils  # noqa: F401
