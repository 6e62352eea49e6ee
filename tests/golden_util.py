"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests."""
import numpy as np

MODEL_NAMES = ["DoubleInt4D", "DoubleInt6D", "Car3D", "Unicycle4D", "Quadcopter6D", "Human6D", "HumanLin6D",
               "Quadcopter12D"]
PASS_CASES = ["cfg1_di4d_k3", "cfg2_di4d_k5", "uni4d_k3", "quad6d_k3", "mixed_q6h6", "di6d_hlin6d_k2", "car3d_k2",
              "quad12d_k2", "di4d_k1"]
MISC_SOLVES = ["cfg1", "uni_k3", "uni_k4", "quad_k3", "quad_k5", "di_k1", "mixed"]
CFG2_SEEDS = [0, 1, 2, 3, 17, 19, 26, 29, 36, 5, 8, 13]


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    den = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b)) / den)


def cfg2_params():
    """Cost/dynamics parameters of scripts/analysis.py:62-69,140-143 for 5 x DoubleInt4D."""
    return dict(model=[0] * 5, n_dims=[2] * 5, Q=np.diag([1.0, 1, 0, 0]), R=np.eye(2), Qf=1000.0 * np.eye(4),
                radius=0.5, dt=0.1, T=50)
