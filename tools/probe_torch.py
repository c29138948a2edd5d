import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device="cuda")
    from visual_odometry_amd import capi
    c = capi.Context(0)
    import numpy as np
    from visual_odometry_amd import synth
    k, d = c.orb_detect(synth.frame_pair(1)[0], 500)
    print("detect ok", len(k))
    y = torch.ones(8, device="cuda"); print(y.sum().item())
else:
    from visual_odometry_amd import capi
    c = capi.Context(0)
    import torch
    print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device="cuda"); print(x.sum().item())
os.system("ls /opt/rocm/lib/libamdhip64.so* ; python -c \"import torch,os; print(os.path.dirname(torch.__file__))\"; ls $(python -c 'import torch,os; print(os.path.dirname(torch.__file__))')/lib | grep -i -E 'hip|rccl' | head")
