"""MvsConfig mirror (TMVS/mvs/mvs.h:19-72) and the two documented parameter sets."""
from __future__ import annotations

from dataclasses import dataclass, asdict

from . import _lib


@dataclass
class MvsConfig:
    cellSize: int = 4
    patchRadius: int = 15
    minCamNum: int = 3
    textureVariation: float = 36.0
    visibleCorrelation: float = 0.7
    minCorrelation: float = 0.7
    maxFitness: float = 10.0
    lodRatio: float = 0.8
    minLOD: int = 0
    maxLOD: int = 15
    maxCellPatchNum: int = 3
    reduceNormalRange: float = 2.0
    adaptiveDistanceEnable: bool = True
    adaptiveDifferenceEnable: bool = True
    adaptiveGradientEnable: bool = False
    distWeighting: float = 5.0
    diffWeighting: float = 128.0 * 128.0
    gradientWeighting: float = 10.0
    neighborRadius: float = 0.005
    neighborRadiusScalar: float = 0.0025
    minRegionRatio: float = 0.55
    depthRangeScalar: float = 1.0
    particleNum: int = 5
    maxIteration: int = 10
    expansionStrategy: int = 0

    @property
    def patchSize(self) -> int:
        return 2 * self.patchRadius + 1

    def to_c(self) -> "_lib.Config":
        c = _lib.Config()
        for k, v in asdict(self).items():
            setattr(c, k, int(v) if isinstance(v, bool) else v)
        c.patchSize = self.patchSize
        return c


def default_config(**over) -> MvsConfig:
    """Compiled-in defaults, TMVS/TMVS.cpp:26-52."""
    c = MvsConfig()
    c.distWeighting = c.patchRadius / 3.0
    for k, v in over.items():
        setattr(c, k, v)
    return c


def readme_config(**over) -> MvsConfig:
    """The documented config.txt, /root/reference/README.md:110-207."""
    c = default_config(
        patchRadius=15, reduceNormalRange=2.0, adaptiveDistanceEnable=True, distWeighting=5.0,
        adaptiveDifferenceEnable=True, diffWeighting=16384.0, adaptiveGradientEnable=False, gradientWeighting=10.0,
        visibleCorrelation=0.7, depthRangeScalar=8.0, particleNum=15, maxIteration=30, cellSize=2,
        maxCellPatchNum=3, expansionStrategy=0, textureVariation=36.0, minLOD=0, maxLOD=15, lodRatio=0.8,
        minCamNum=3, minCorrelation=0.9, minRegionRatio=0.15, maxFitness=10.0, neighborRadiusScalar=0.01)
    for k, v in over.items():
        setattr(c, k, v)
    return c
