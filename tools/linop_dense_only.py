"""The dense 1024 x 1024 operator on 16 384 rows, matrix-core route, 60 launches (for counter passes: tools/linop_clock.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch      # noqa: E402
from cosmoprimo_amd.spline import LinearOperator      # noqa: E402

dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
op = LinearOperator.dense(rng.normal(size=(1024, 1024)), device=dev)
y = torch.as_tensor(rng.normal(size=(16384, 1024)), device=dev)
for _ in range(60):
    op(y, path='mfma')
torch.cuda.synchronize()
