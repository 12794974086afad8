"""Diagnostic: run a tool with the process bound to the CPUs next to the GPU, to the other socket's, or unbound
(usage: numa_ab.py local|remote|any script.py [args...]) -- what a two-socket host's thread placement does to a host-side figure."""
import os, sys, runpy
which, script = sys.argv[1], sys.argv[2]
import torch
p = torch.cuda.get_device_properties(0)
path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/local_cpulist" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
local = set()
for part in open(path).read().strip().split(","):
    lo, _, hi = part.partition("-")
    local |= set(range(int(lo), int(hi or lo) + 1))
allc = set(range(os.cpu_count()))
if which == "local": os.sched_setaffinity(0, local)
elif which == "remote" and allc - local: os.sched_setaffinity(0, allc - local)
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
