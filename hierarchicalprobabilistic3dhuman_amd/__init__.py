"""MI355X-native per-image inference path of HierarchicalProbabilistic3DHuman.

Host-side mirror of the reference interface for that path (same names, argument meaning and error
behaviour) over libhps.so, a C-ABI library of hand-written gfx950 HIP kernels (include/hps.h):

    reference module                              here
    models/poseMF_shapeGaussian_net.py       ->   poseMF_shapeGaussian_net.PoseMFShapeGaussianNet
    models/resnet.py                         ->   resnet.resnet18
    models/smpl_official.py                  ->   smpl_official.SMPL
    utils/sampling_utils.py                  ->   sampling_utils.*
    utils/rigid_transform_utils.py           ->   rigid_transform_utils.*
    predict/predict_poseMF_shapeGaussian_net ->   predict_poseMF_shapeGaussian_net.*
    models/canny_edge_detector.py            ->   canny_edge_detector.CannyEdgeDetector
    utils/label_conversions.py (heat-maps)   ->   label_conversions.*
"""
from .configs import get_cfg_defaults, SMPL_PARENTS  # noqa: F401

__version__ = "0.1.0"
