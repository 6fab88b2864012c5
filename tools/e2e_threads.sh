#!/bin/bash
# tools/e2e_threads.sh -- end_to_end / latency figures of bench.py for library variants built with different POLAR_HOST_THREADS
for v in "$@"; do
  lib=polardecoding_amd/lib/libpolar_hip.so
  [ "$v" != base ] && lib=build/variants/libpolar_hip_$v.so
  python tools/run_with_lib.py $lib bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fer-sweep --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']
print('$v', 'e2e %.3f M frames/s  %.1f GB/s  %.2f ms' % (e['value']/1e6, e['pcie_GBps'], e['seconds']*1e3), ' latency %.1f us' % d['single_frame_latency_us']['median'])"
done
