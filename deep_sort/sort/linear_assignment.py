from yolo_deepsort_amd.sort_api import (INFTY_COST, gate_cost_matrix, linear_assignment, matching_cascade,  # noqa: F401
                                        min_cost_matching)
