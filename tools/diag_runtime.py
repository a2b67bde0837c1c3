import os, sys, ctypes
mode = sys.argv[1]
print("== mode", mode, {k: v for k, v in os.environ.items() if k.startswith(("HIP", "HSA", "ROC", "GPU", "CUDA"))})
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l))
if mode == "torch":
    import torch
    print("is_available", torch.cuda.is_available(), "count", torch.cuda.device_count())
    x = torch.zeros(4, device="cuda"); print("torch alloc ok", x.sum().item())
elif mode == "lib":
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nncf_amd", "libnncf_hip.so"))
    print("lib count", L.nncf_device_count())
elif mode == "torch_lib":
    import torch
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nncf_amd", "libnncf_hip.so"))
    print("lib count (before torch init)", L.nncf_device_count())
    x = torch.zeros(4, device="cuda"); print("torch alloc ok", x.sum().item())
    print("lib count (after torch init)", L.nncf_device_count())
elif mode == "torchinit_lib":
    import torch
    x = torch.zeros(4, device="cuda"); print("torch alloc ok", x.sum().item())
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nncf_amd", "libnncf_hip.so"))
    print("lib count (after torch init)", L.nncf_device_count())
print(maps())
