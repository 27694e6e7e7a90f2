import sys, os
sys.path.insert(0, ".")
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.ones(4, device="cuda:0"); print(x.sum().item())
    from horayzon_amd import _lib
    print("hz devices", _lib.device_count(), _lib.device_info(0))
elif order == "hz_first":
    from horayzon_amd import _lib
    print("hz devices", _lib.device_count(), _lib.device_info(0))
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.ones(4, device="cuda:0"); print(x.sum().item())
print(maps())
