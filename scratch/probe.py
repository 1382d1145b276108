import torch, ctypes, os, numpy as np, subprocess
print("torch", torch.__version__, "hip", torch.version.hip, "cuda avail", torch.cuda.is_available())
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
print("rt version seen by lib", lib.probe_rtver())
maps = open("/proc/self/maps").read()
print("hip libs mapped:", sorted({l.split()[-1] for l in maps.splitlines() if "amdhip" in l}))
x = torch.randn(1000, device="cuda"); y = torch.randn(1000, device="cuda"); y0 = y.clone()
s = torch.cuda.current_stream().cuda_stream
lib.probe_axpy.argtypes=[ctypes.c_void_p,ctypes.c_void_p,ctypes.c_float,ctypes.c_int,ctypes.c_void_p]
r = lib.probe_axpy(x.data_ptr(), y.data_ptr(), 2.0, 1000, s); torch.cuda.synchronize()
print("axpy rc", r, "err", (y-(2*x+y0)).abs().max().item())
K=64
A = torch.randn(32,K,device="cuda"); B = torch.randn(K,32,device="cuda"); C = torch.zeros(32,32,device="cuda")
lib.probe_mfma.argtypes=[ctypes.c_void_p]*3+[ctypes.c_int,ctypes.c_void_p]
r = lib.probe_mfma(A.data_ptr(),B.data_ptr(),C.data_ptr(),K,s); torch.cuda.synchronize()
ref = (A.double()@B.double()).float()
print("mfma rc", r, "err", (C-ref).abs().max().item())
# side stream
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    r = lib.probe_axpy(x.data_ptr(), y.data_ptr(), 1.0, 1000, torch.cuda.current_stream().cuda_stream)
st.synchronize(); print("side stream ok", (y-(3*x+y0)).abs().max().item())
# graph capture
g = torch.cuda.CUDAGraph()
yy = torch.zeros(1000, device="cuda")
with torch.cuda.graph(g):
    lib.probe_axpy(x.data_ptr(), yy.data_ptr(), 1.0, 1000, torch.cuda.current_stream().cuda_stream)
g.replay(); g.replay(); torch.cuda.synchronize()
print("graph replay", (yy-2*x).abs().max().item())
print(subprocess.run("nproc; lscpu | grep 'Model name'; free -g | head -2; rocm-smi --showmeminfo vram | head -8", shell=True, capture_output=True, text=True).stdout)
