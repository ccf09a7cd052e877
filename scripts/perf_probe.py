"""is perf_event_open usable on this box?  (process-wide hardware counters with inherit: a perf-stat-style reading for bench.py --diag)"""
import ctypes, os, struct, threading, time

libc = ctypes.CDLL(None, use_errno=True)
PERF_TYPE_HARDWARE, PERF_TYPE_HW_CACHE = 0, 3
HW = {"cycles": 0, "instructions": 1, "cache_references": 2, "cache_misses": 3, "stalled_backend": 8}


def open_counter(config, typ=PERF_TYPE_HARDWARE, inherit=True):
    # struct perf_event_attr (first 112 bytes are enough: size field says so)
    attr = bytearray(128)
    struct.pack_into("IIQQQQ", attr, 0, typ, 128, config, 0, 0, 0)  # type, size, config, sample_period, sample_type, read_format
    flags = (1 << 0) | ((1 << 1) if inherit else 0) | (1 << 5) | (1 << 6)  # disabled | inherit | exclude_kernel | exclude_hv
    struct.pack_into("Q", attr, 40, flags)
    buf = (ctypes.c_char * 128).from_buffer(attr)
    fd = libc.syscall(298, buf, 0, -1, -1, 0)
    if fd < 0:
        raise OSError(ctypes.get_errno(), os.strerror(ctypes.get_errno()))
    return fd


print("perf_event_paranoid:", open("/proc/sys/kernel/perf_event_paranoid").read().strip() if os.path.exists("/proc/sys/kernel/perf_event_paranoid") else "?")
fds = {}
for k, c in HW.items():
    try:
        fds[k] = open_counter(c)
    except OSError as e:
        print(k, "->", e)
PERF_EVENT_IOC_ENABLE, PERF_EVENT_IOC_DISABLE = 0x2400, 0x2401
for fd in fds.values():
    libc.ioctl(fd, PERF_EVENT_IOC_ENABLE, 0)


def spin():
    t = time.time()
    x = 0
    while time.time() - t < 0.2:
        x += 1


th = [threading.Thread(target=spin) for _ in range(2)]
[t.start() for t in th]
[t.join() for t in th]
for k, fd in fds.items():
    libc.ioctl(fd, PERF_EVENT_IOC_DISABLE, 0)
    print(k, struct.unpack("Q", os.read(fd, 8))[0])
