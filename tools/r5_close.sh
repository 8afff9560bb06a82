#!/bin/bash
# closing runs of the round on one box: gradient-yardstick sweep, NaN-poisoned arena suite, 3000-step trainer run, three default bench lines,
# the other workloads
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; out=gpurun_out/r5/close; mkdir -p $out
timeout 600 python3 tests/grad_yardstick_sweep.py --draws 16 --base-draws 4 > $out/grad_yardstick.txt 2> $out/grad_yardstick.err < /dev/null; tail -3 $out/grad_yardstick.txt
WJ_ARENA_FILL=nan timeout 1500 python3 -m pytest tests -m gpu -q > $out/gputest_nan_arena.log 2>&1 < /dev/null; tail -2 $out/gputest_nan_arena.log
timeout 900 python3 train.py trainer.steps=3000 trainer.warmup_steps=500 trainer.log_every_n_steps=250 > $out/train_3000steps.log 2>&1 < /dev/null; tail -3 $out/train_3000steps.log
for i in 1 2 3; do timeout 300 python3 bench.py --no-cpu-baseline --dense-steps 0 > $out/bench_$i.json 2> $out/bench_$i.err < /dev/null; tail -1 $out/bench_$i.json | cut -c1-170; done
for w in 4s-bf16 4s-fp8 2s-nat; do timeout 400 python3 bench.py --workload $w --no-cpu-baseline --dense-steps 0 --no-profile > $out/bench_$w.json 2> $out/bench_$w.err < /dev/null; tail -1 $out/bench_$w.json | cut -c1-170; done
