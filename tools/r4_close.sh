#!/bin/bash
# closing runs of the round on one box: NaN-poisoned arena suite, 3000-step trainer run, three default bench lines
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$root"; out=gpurun_out/r4/close; mkdir -p $out
WJ_ARENA_FILL=nan timeout 1500 python3 -m pytest tests -m gpu -q > $out/gputest_nan_arena.log 2>&1; tail -2 $out/gputest_nan_arena.log
timeout 900 python3 train.py trainer.steps=3000 trainer.warmup_steps=500 trainer.log_every_n_steps=250 > $out/train_3000steps.log 2>&1; tail -3 $out/train_3000steps.log
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --dense-steps 0 > $out/bench_$i.json 2> $out/bench_$i.err; tail -1 $out/bench_$i.json | cut -c1-170; done
