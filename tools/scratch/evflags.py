import ctypes as C
hip = C.CDLL("libamdhip64.so")
for fl in (0x2, 0x2|0x20000000, 0x2|0x40000000, 0x2|0x20000000|0x40000000, 0x20000000, 0x40000000):
    e = C.c_void_p()
    r = hip.hipEventCreateWithFlags(C.byref(e), C.c_uint(fl))
    print(hex(fl), "create ->", r)
    if r == 0:
        print("   record ->", hip.hipEventRecord(e, None), "sync ->", hip.hipEventSynchronize(e))
