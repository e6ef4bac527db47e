import ctypes, os, sys, subprocess
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'hip' in l or 'hsa' in l))
mode = sys.argv[1]
print("MODE", mode, {k:v for k,v in os.environ.items() if 'HIP' in k or 'ROC' in k or 'HSA' in k or 'CUDA' in k})
if mode == "sys":
    l = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so'); n = ctypes.c_int(-1)
    print(l.hipGetDeviceCount(ctypes.byref(n)), n.value); print(maps())
elif mode == "gpsiq_first":
    sys.path.insert(0, 'multi-sdr-gps-sim_amd'); import gpsiq
    try: c = gpsiq.Context(0); print("ctx ok")
    except Exception as e: print("ERR", e)
    print(maps())
elif mode == "torch_noinit":
    import torch; sys.path.insert(0, 'multi-sdr-gps-sim_amd'); import gpsiq
    print(torch.cuda.is_available())
    try: c = gpsiq.Context(0); print("ctx ok")
    except Exception as e: print("ERR", e)
    print(maps())
elif mode == "torch_init":
    import torch; sys.path.insert(0, 'multi-sdr-gps-sim_amd'); import gpsiq
    torch.cuda.init(); print(torch.cuda.device_count())
    try: c = gpsiq.Context(0); print("ctx ok")
    except Exception as e: print("ERR", e)
    print(maps())
elif mode == "gpsiq_then_torch":
    sys.path.insert(0, 'multi-sdr-gps-sim_amd'); import gpsiq
    print(maps())
    import torch
    print(maps())
    print("avail", torch.cuda.is_available())
    try: c = gpsiq.Context(0); print("ctx ok")
    except Exception as e: print("ERR", e)
    try: torch.cuda.init(); print("torch init ok", torch.zeros(3, device='cuda').sum().item())
    except Exception as e: print("TORCH ERR", e)
elif mode == "gpsiq_ctx_then_torch":
    sys.path.insert(0, 'multi-sdr-gps-sim_amd'); import gpsiq
    c = gpsiq.Context(0); print("ctx ok")
    import torch
    print("avail", torch.cuda.is_available())
    try: print("torch ok", torch.zeros(3, device='cuda').sum().item())
    except Exception as e: print("TORCH ERR", e)
    print(maps())
