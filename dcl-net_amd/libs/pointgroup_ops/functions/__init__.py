from . import pointgroup_ops  # noqa: F401
