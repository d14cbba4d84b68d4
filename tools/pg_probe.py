import sys, numpy as np, torch
sys.path.insert(0, '.')
from ceracoder_amd import enc as E, synth
def trial(w, h, depth, sample, n=200, gop=60, dev=True):
    clip = list(synth.s2_frames(w, h, 8))
    e = E.Encoder(w, h, fps=60, gop=gop, bitrate_bps=6_000_000, pipeline_depth=depth, profile_events=sample, exclusive=True)
    bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
    torch.cuda.synchronize()
    try:
        for i in range(n):
            if dev:
                p = bufs[i % 8].data_ptr(); e.submit_device(p, w, p + w * h, w, pts=i)
            else:
                e.submit(*clip[i % 8], pts=i)
            if e.pending > depth: e.collect(copy=False)
        while e.pending: e.collect(copy=False)
        print("ok  ", w, h, depth, sample, dev, flush=True)
    except Exception as ex:
        print("FAIL", w, h, depth, sample, dev, ex, flush=True)
    try: e.close()
    except Exception: pass
trial(*[int(x) for x in sys.argv[1:5]], dev=bool(int(sys.argv[5])))
