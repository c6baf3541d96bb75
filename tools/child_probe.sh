# development: is the unchanged caller's bimodal step time (0.362 / 0.426 ms per process) thread placement?
BDF=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(f"{getattr(p,'pci_domain_id',0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0")
PY
)
NODECPUS=$(cat /sys/bus/pci/devices/$BDF/local_cpulist)
NODE=$(cat /sys/bus/pci/devices/$BDF/numa_node)
FIRST=$(echo $NODECPUS | cut -d- -f1 | cut -d, -f1)
L3=$(cat /sys/devices/system/cpu/cpu$FIRST/cache/index3/shared_cpu_list)
OTHER=$(( NODE == 0 ? 64 : 0 ))
echo "gpu $BDF node $NODE cpus $NODECPUS ; L3 group of cpu$FIRST: $L3 ; a cpu of the other socket: $OTHER"
run() { tag=$1; shift; "$@" python bench.py --gpus 1 --no-cpu-baseline --no-secondary --min-seconds 1.0 --min-trials 5 --config C2 --api module 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d['trials_ms']; print('$tag', d['value'], d['ms_per_step'], 'min', min(t), 'max', max(t))"; }
for i in 1 2 3 4 5 6; do run unbound env SPF_BIND=0; done
for i in 1 2 3 4 5 6; do run bound_by_bench env; done
for i in 1 2; do run node env SPF_BIND=0 taskset -c $NODECPUS; done
for i in 1 2; do run l3group env SPF_BIND=0 taskset -c $L3; done
for i in 1 2; do run two_sockets env SPF_BIND=0 taskset -c $FIRST,$OTHER; done
