#!/bin/bash
cd $GRAFT_REPO_ROOT
DA_WG16=1 python tools/ab/w16_compare.py /tmp/w16_on.npz 2>&1 | grep -v amdgpu.ids
DA_WG16=0 python tools/ab/w16_compare.py /tmp/w16_off.npz 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import numpy as np
a, b = np.load('/tmp/w16_on.npz'), np.load('/tmp/w16_off.npz')
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    print(k, 'rel-l2 %.3e' % (np.linalg.norm(x - y) / np.linalg.norm(y)), 'max-abs / max %.3e' % (np.abs(x - y).max() / np.abs(y).max()), 'finite', bool(np.isfinite(x).all()))
PY
