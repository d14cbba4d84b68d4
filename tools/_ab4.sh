set -e
mkdir -p gpurun_out/ab
run() { n=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-gst-latency --steps 480 > gpurun_out/ab/bench_$n.log 2>&1
  grep '^{' gpurun_out/ab/bench_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_picture']; print('$n', d['value'], 'me',s['me'],'sel',s['me_select_x3'],'p',s['p_stage_total'],'db',s['deblock_wavefront'],'lat',d['latency_ms']['p50'])"
}
run base A=1
run f192 MI355ENC_FRONT_CUS=192
run f128 MI355ENC_FRONT_CUS=128
run f192p MI355ENC_FRONT_CUS=192 MI355ENC_FRONT_PAT=1
run f128p MI355ENC_FRONT_CUS=128 MI355ENC_FRONT_PAT=1
run base2 A=1
