for b in 0 1 2; do
  MFK_SCAN_DBG=$b timeout -k 10 120 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile gpurun_out/prof_d$b.txt > /dev/null 2>&1 || exit 1
  echo "dbg $b: $(grep k_scan_clips gpurun_out/prof_d$b.txt) | $(tail -1 gpurun_out/prof_d$b.txt)"
done
