#!/bin/bash
# Soak of the final binary: bench.py's default configuration and config 4 over 20 000 steps of every form (counters must stay clean),
# and 20 000 control ticks through ndp_tick.  gpurun --timeout 1800 -- 'bash scripts/soak.sh'  ->  gpurun_out/soak/soak.txt
O=$PWD/gpurun_out/soak; mkdir -p $O
sm() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 %.2f M solves/s, %.2f us/step, parity %s, not converged %s watchdog: %s' % (d['value']/1e6 if d['value'] else -1, d['ms_per_step']*1e3, d.get('parity_max_rel_vs_oracle'), d.get('not_converged'), d.get('watchdog')))
for blk in ('forms','downwash_forms','scaling_baseline','exchange'):
    for k,v in (d.get(blk) or {}).items():
        if isinstance(v, dict): print('    %s %s %s' % (blk, k, {kk: vv for kk, vv in v.items() if kk in ('value','us','ok','counters','ticks','ack_timeouts','epoch_timeouts','slot_mismatches','force_timeouts','slot_timeouts')}))
"; }
timeout 900 python3 bench.py --steps 20000 --warmup 50 2> $O/default.err | sm soak_default | tee $O/soak.txt
timeout 600 python3 bench.py --config 4 --formations 512 --steps 20000 --warmup 50 2> $O/c4.err | sm soak_c4 | tee -a $O/soak.txt
timeout 900 python3 scripts/tick_soak.py 2> $O/tick.err | grep -v amdgpu | tee -a $O/soak.txt
