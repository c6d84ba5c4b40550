import sys, os, torch
sys.path.insert(0, os.getcwd())
print("priority range", torch.cuda.Stream.priority_range())
import ctypes as C
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
print("hipDeviceGetStreamPriorityRange rc", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "least", lo.value, "greatest", hi.value)
