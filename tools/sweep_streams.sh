for S in 2 3 4 5 6 8; do echo -n "streams=$S: "; python3 bench.py --no-cpu-baseline --steps 24 --warmup 8 --streams $S 2>&1 >/dev/null | grep "bench\] gpu" | cut -c1-70; done
for C in 2 4; do echo -n "tune_concurrency=$C streams=4: "; OVC_TUNE_CONCURRENCY=$C python3 bench.py --no-cpu-baseline --steps 24 --warmup 8 --streams 4 2>&1 >/dev/null | grep "bench\] gpu" | cut -c1-70; done
for B in 512 1024; do echo -n "batch=$B streams=4: "; python3 bench.py --no-cpu-baseline --steps 12 --warmup 4 --streams 4 --batch $B 2>&1 >/dev/null | grep "bench\] gpu" | cut -c1-70; done
echo -n "batch=128 streams=8: "; python3 bench.py --no-cpu-baseline --steps 48 --warmup 16 --streams 8 --batch 128 2>&1 >/dev/null | grep "bench\] gpu" | cut -c1-70
