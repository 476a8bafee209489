"""Darknet .cfg parsing, restated from reference yolo3/utils/parse_config.py:1-19."""


def parse_model_config_text(text):
    """Same block/key/value semantics as the reference parser: blank and '#'
    lines dropped, whitespace stripped, a dict per [section]; 'convolutional'
    gets batch_normalize=0 (int) by default; values from the file stay str."""
    lines = text.split("\n")
    lines = [x for x in lines if x and not x.startswith("#")]
    lines = [x.rstrip().lstrip() for x in lines]
    defs = []
    for line in lines:
        if line.startswith("["):
            defs.append({})
            defs[-1]["type"] = line[1:-1].rstrip()
            if defs[-1]["type"] == "convolutional":
                defs[-1]["batch_normalize"] = 0
        else:
            key, value = line.split("=")
            defs[-1][key.rstrip()] = value.strip()
    return defs


def parse_model_config(path):
    with open(path, "r") as f:
        return parse_model_config_text(f.read())
