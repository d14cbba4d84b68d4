R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pl -o pl -- python3 $R/bench.py --steps 120 --warmup 20 --no-gst-latency --no-cpu-baseline --overlap 4 --cavlc-threads 4 --sample 1000 > $R/gpurun_out/prof_pl.log 2>&1
cd $R
f=$(find gpurun_out/prof_pl -name "*.db" | head -1)
python3 tools/rocpd_summary.py $f > gpurun_out/prof_pl_summary.txt 2>&1
tail -80 gpurun_out/prof_pl_summary.txt
