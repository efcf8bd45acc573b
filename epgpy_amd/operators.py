"""operator namespace (mirrors epgpy/operators.py:1-25, hot-path subset)"""
from .operator import (Operator, MultiOperator, EmptyOperator, Spoiler, Wait, Offset, Reset, PD,
                       NULL, SPOILER, RESET)
from .probe import Probe, Adc, ADC
from .opmatrix import MatrixOp
from .opscalar import ScalarOp
from .evolution import E, P, R
from .transition import T, Tx, Ty, Phi
from .shift import S
from .diffusion import D
from .diff import Jacobian, Hessian, PartialsPruner
