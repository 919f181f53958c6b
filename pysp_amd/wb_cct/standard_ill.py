"""Standard-illuminant chromaticities used by the colour path (reference wb_cct/standard_ill.py:27-40)."""
from enum import IntEnum, auto
from typing import Dict, Tuple


class StandardIlluminant(IntEnum):
    A = auto()
    B = auto()
    C = auto()
    D50 = auto()
    D55 = auto()
    D65 = auto()
    D75 = auto()


_XY: Dict[StandardIlluminant, Tuple[float, float]] = {
    StandardIlluminant.A: (0.44758, 0.40745),
    StandardIlluminant.B: (0.34842, 0.35161),
    StandardIlluminant.C: (0.31006, 0.31616),
    StandardIlluminant.D50: (0.34567, 0.35850),
    StandardIlluminant.D55: (0.33242, 0.34743),
    StandardIlluminant.D65: (0.31272, 0.32903),
    StandardIlluminant.D75: (0.29902, 0.31485),
}


def get_chromacity_from_illuminant(illuminant: StandardIlluminant) -> Tuple[float, float]:
    return _XY[illuminant]
