for ov in 2 3 4 5; do timeout -k 10 100 python bench.py --no-gst-latency --no-cpu-baseline --overlap $ov --cavlc-threads 4 --sample 30 > gpurun_out/b_ov${ov}.log 2>&1; done
grep -h -o '"value": [0-9.]*' gpurun_out/b_ov2.log gpurun_out/b_ov3.log gpurun_out/b_ov4.log gpurun_out/b_ov5.log
