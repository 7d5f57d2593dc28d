"""Which HIP / HSA / RCCL copies does a process end up with?  (PyTorch ships its own next to /opt/rocm's.)  `plain`: this library first,
then torch; `torch_first`: the other way round.  comm.hip loads the librccl that sits beside the libamdhip64 it is bound to -- asked for
by soname, dlopen would hand back torch's copy, bound to a runtime that was never initialised ("no ROCm-capable device").
usage: python tools/diag_rccl.py plain|torch_first"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
def maps(tag):
    libs = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if any(s in ln for s in ("rccl", "amdhip", "hsa-runtime"))})
    print(tag, libs, flush=True)
if mode == "torch_first":
    import torch; torch.cuda.init(); print("torch cuda ok", torch.cuda.device_count())
import mendeliht_amd as m
print("devices", m.device_count())
maps("after our lib")
import torch
maps("after import torch")
from mendeliht_amd import api
uid = (C.c_char * 128)()
rc = api.lib().mih_rccl_unique_id(uid)
print("unique id rc", rc)
maps("after rccl_load")
h = C.c_void_p(None)
rc = api.lib().mih_comm_create_rccl(uid, 0, 1, 0, 0, 10, C.byref(h))
buf = C.create_string_buffer(512); api.lib().mih_last_error(buf, 512)
print("create rc", rc, buf.value.decode()[:200])
