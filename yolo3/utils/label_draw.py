from yolo_deepsort_amd.label_draw import *  # noqa: F401,F403  (import-path shim, reference yolo3/utils/label_draw.py)
from yolo_deepsort_amd.label_draw import LabelDrawer, draw_rects, draw_rects_and_labels  # noqa: F401
