import sys
sys.path.insert(0, "/root/repo")
from ceracoder_amd import enc as E, synth
for i4 in (True, False):
    for qp in (24, 40):
        e = E.Encoder(1920, 1080, gop=60, fixed_qp=qp, i4x4=i4)
        fr = list(synth.s2_frames(1920, 1080, 1))
        e.encode(*fr[0])
        t = e.time_stage(E.STAGE_INTRA, 5)
        mbi = e.fetch(E.FETCH_MBINFO)
        print("i4x4=%s qp=%d: intra %.3f ms (%.2f us/step), I4 macroblocks %.0f%%" % (i4, qp, t, t * 1e3 / 187, 100.0 * (mbi["mb_type"] == 2).mean()), flush=True)
        e.close()
