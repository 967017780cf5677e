import anakin.criterions  # noqa: F401  (populate the registries, as the reference's package __init__s do)
import anakin.datasets  # noqa: F401
import anakin.metrics  # noqa: F401
import anakin.models  # noqa: F401
from artiboost_amd.registry import (BACKBONE, DATASET, HEAD, LOSS, METRIC, MODEL, NECK, build, build_arch_model_list,  # noqa: F401
                                    build_backbone, build_criterion_loss_list, build_dataset, build_evaluator_metric_list,
                                    build_head, build_loss, build_metric, build_model)
