mkdir -p gpurun_out/r4x
timeout -k 10 120 python tools/probe_me.py > gpurun_out/r4x/me.log 2>&1; cat gpurun_out/r4x/me.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "me_ or select or motion or pslices or stream" > gpurun_out/r4x/t.log 2>&1; tail -3 gpurun_out/r4x/t.log
