"""celeste.jl_amd -- MI355X-native ELBO evaluation engine for Celeste (hot path only).

Import as ``celeste_jl_amd`` (see celeste_jl_amd.py at the repository root).
"""
from . import cabi, params, model  # noqa: F401
from .params import (ids, CatalogEntry, generic_init_source, catalog_init_source, init_sources,  # noqa: F401
                     perturb_params)
from .elbo import (ElboArgs, ElboConfig, SensitiveFloat, FieldContext, elbo, elbo_likelihood,  # noqa: F401
                   maximize)

from .infer import (BoundingBox, OptimizedSource, infer_box, one_node_single_infer,  # noqa: F401
                    one_node_joint_infer)

__all__ = ["BoundingBox", "OptimizedSource", "infer_box", "one_node_single_infer", "one_node_joint_infer", "ElboArgs", "ElboConfig", "maximize", "SensitiveFloat", "FieldContext", "elbo", "elbo_likelihood", "ids", "CatalogEntry",
           "generic_init_source", "catalog_init_source", "init_sources", "perturb_params"]
