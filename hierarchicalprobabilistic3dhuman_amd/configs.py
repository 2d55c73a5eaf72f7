"""Configuration values the hot path reads.

The reference keeps these in a yacs ``CfgNode`` (configs/poseMF_shapeGaussian_net_config.py:4-24);
the hot path only ever reads attributes, so any attribute bag with the same names works.  yacs
objects are accepted unchanged wherever a ``config`` is taken.
"""
from types import SimpleNamespace

# SMPL kinematic tree (public constant; the reference reads it from the SMPL pkl at run_predict.py:65).
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]

NUM_VERTS = 6890
NUM_JOINTS = 24

# smplx VertexJointSelector: 21 vertices appended after the 24 kinematic joints
# (face, feet, left-hand tips, right-hand tips) -- SURVEY.md section 8(a) note.
SMPLX_EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583,
                          3216, 3226, 3387, 6617, 6624, 6787,
                          2746, 2319, 2445, 2556, 2673,
                          6191, 5782, 5905, 6016, 6133]


def get_cfg_defaults():
    """Values of configs/poseMF_shapeGaussian_net_config.py:8-24 that the inference path reads."""
    return SimpleNamespace(
        MODEL=SimpleNamespace(NUM_IN_CHANNELS=18, NUM_RESNET_LAYERS=18, EMBED_DIM=256,
                              DELTA_I=True, DELTA_I_WEIGHT=1.0, NUM_SMPL_BETAS=10),
        DATA=SimpleNamespace(PROXY_REP_SIZE=256, HEATMAP_GAUSSIAN_STD=4.0, EDGE_NMS=True,
                             EDGE_THRESHOLD=0.0, EDGE_GAUSSIAN_STD=1.0, EDGE_GAUSSIAN_SIZE=5,
                             BBOX_THRESHOLD=0.95, BBOX_SCALE_FACTOR=1.2),
    )
