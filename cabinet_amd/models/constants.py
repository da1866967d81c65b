"""Model constants (mirror of reference src/models/constants.py:6-33; values only)."""

MOBILENET_LARGE_FEATURES = 960
MOBILENET_SMALL_FEATURES = 576

MODEL_CONFIG = {
    "large": {"attention_planes": MOBILENET_LARGE_FEATURES, "output_channel": 1280},
    "small": {"attention_planes": MOBILENET_SMALL_FEATURES, "output_channel": 1024},
}

OHEM_DIVISOR = 16
DEFAULT_SCORE_THRESHOLD = 0.7
EVAL_STRIDE_RATE = 5 / 6.0
DEFAULT_EVAL_SCALES = [0.5, 0.75, 1.0, 1.25, 1.5, 1.75]
CITYSCAPES_NUM_CLASSES = 19
UAVID_NUM_CLASSES = 8
DEFAULT_IGNORE_LABEL = 255
VISUALIZATION_SAMPLE_LIMIT = 50

# backbone tables (reference configs/model/mobilenetv3_{large,small}.yaml): k, t, c, SE, HS, s
MOBILENETV3_CFGS = {
    "large": [
        [3, 1, 16, 0, 0, 1], [3, 4, 24, 0, 0, 2], [3, 3, 24, 0, 0, 1],
        [5, 3, 40, 1, 0, 2], [5, 3, 40, 1, 0, 1], [5, 3, 40, 1, 0, 1],
        [3, 6, 80, 0, 1, 2], [3, 2.5, 80, 0, 1, 1], [3, 2.3, 80, 0, 1, 1],
        [3, 2.3, 80, 0, 1, 1], [3, 6, 112, 1, 1, 1], [3, 6, 112, 1, 1, 1],
        [5, 6, 160, 1, 1, 2], [5, 6, 160, 1, 1, 1], [5, 6, 160, 1, 1, 1],
    ],
    "small": [
        [3, 1, 16, 1, 0, 2], [3, 4.5, 24, 0, 0, 2], [3, 3.67, 24, 0, 0, 1],
        [5, 4, 40, 1, 1, 2], [5, 6, 40, 1, 1, 1], [5, 6, 40, 1, 1, 1],
        [5, 3, 48, 1, 1, 1], [5, 3, 48, 1, 1, 1], [5, 6, 96, 1, 1, 2],
        [5, 6, 96, 1, 1, 1], [5, 6, 96, 1, 1, 1],
    ],
}
