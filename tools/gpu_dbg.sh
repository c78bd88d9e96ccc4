#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "--- lib first, then torch"
RSX_VERBOSE=1 python - <<'PY'
import ctypes
from radix_sort_amd import _lib
L = _lib.load()
h = ctypes.c_void_p()
print("create rc (before torch)", L.rsx_ctx_create(0, ctypes.byref(h)))
import torch
print(torch.cuda.is_available())
h2 = ctypes.c_void_p()
print("create rc (after torch)", L.rsx_ctx_create(0, ctypes.byref(h2)))
PY
echo "--- smoke only"
RSX_VERBOSE=1 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
ldd radix_sort_amd/lib/librsx.so | grep -i hip
python -c "import torch,os; print(os.path.dirname(torch.__file__))"; ls $(python -c "import torch,os; print(os.path.dirname(torch.__file__))")/lib | grep -i amdhip
