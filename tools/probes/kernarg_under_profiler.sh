cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  export HIP_FORCE_DEV_KERNARG=$v
  rm -rf /tmp/kp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$v -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --single-window > /tmp/kp_$v.log 2>&1
  f=$(find /tmp/kp_$v -name '*kernel_stats.csv' | head -1)
  echo "HIP_FORCE_DEV_KERNARG=$v: $(grep cs_kernel $f | awk -F, '{print $(NF-6), $(NF-4)}' | head -1)"
done
