from yolo_deepsort_amd.action import Action, TakeOff, Landing, Glide, FastCrossing, BreakInto  # noqa: F401
