import ctypes, os, glob, torch
libc = ctypes.CDLL(None, use_errno=True)
print("nodes:", sorted(os.path.basename(p) for p in glob.glob("/sys/devices/system/node/node*")))
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
try: print("gpu", bdf, "numa_node", open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip(), "local_cpulist", open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip())
except Exception as e: print("no sysfs for gpu", e)
print(open("/proc/self/status").read().split("Cpus_allowed_list")[1].splitlines()[0], [l for l in open("/proc/self/status").read().splitlines() if "Mems_allowed_list" in l])
for n in sorted(glob.glob("/sys/devices/system/node/node*")): print(os.path.basename(n), "cpus", open(n + "/cpulist").read().strip())
SYS_set_mempolicy, SYS_get_mempolicy = 238, 239
mask = ctypes.c_ulong(1)
r = libc.syscall(SYS_set_mempolicy, 1, ctypes.byref(mask), 64)   # MPOL_PREFERRED node 0
print("set_mempolicy(PREFERRED, node0) ->", r, os.strerror(ctypes.get_errno()) if r else "ok")
r = libc.syscall(SYS_set_mempolicy, 0, None, 0)
print("set_mempolicy(DEFAULT) ->", r)
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print(e)
