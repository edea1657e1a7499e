import sys, torch, numpy as np
sys.path.insert(0, '.')
from diart_amd import models as M, _lib
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream
gpu = torch.device('cuda', 0)
stream = torch.from_numpy(synth_stream(3, 12.0)).to(gpu)
for precision in ('f16x3', 'f32'):
    seg = M.HipSegmentation(synth_segmentation_state(), max_batch=8, precision=precision).to(gpu)
    emb = M.HipEmbedding(synth_embedding_state(), max_batch=8, precision=precision).to(gpu)
    base = stream.unfold(0, 80000, 8000)[:4]
    ref = seg(base[:, None, :].contiguous()).cpu()
    refe = emb(base[:, None, :].contiguous()).cpu()
    for off, hop in ((1, 8000), (2, 8000), (3, 8001), (0, 8003), (1, 7999)):
        v = stream[off:].unfold(0, 80000, hop)[:4]
        want = seg(v.contiguous()[:, None, :]).cpu(); wante = emb(v.contiguous()[:, None, :]).cpu()
        try:
            got = seg(v[:, None, :]).cpu(); gote = emb(v[:, None, :]).cpu()
            print(precision, 'offset', off, 'hop', hop, 'ptr%16', v.data_ptr() % 16, 'seg', (got - want).abs().max().item(), 'emb', (gote - wante).abs().max().item())
        except Exception as e:
            print(precision, 'offset', off, 'hop', hop, 'ERR', type(e).__name__, str(e)[:150])
    # NaN / Inf input
    for bad in (float('nan'), float('inf'), 1e30):
        x = base[:, None, :].contiguous().clone(); x[1, 0, 500] = bad
        try:
            got = seg(x).cpu(); gote = emb(x).cpu()
            try:
                _lib.range_check(gpu.index); flag = 'no flag'
            except Exception as e:
                flag = 'flag: ' + str(e)[:80]
            print(precision, 'input', bad, 'seg row1 finite', torch.isfinite(got[1]).all().item(), 'other rows equal', torch.equal(got[[0,2,3]], ref[[0,2,3]]), 'emb row1 finite', torch.isfinite(gote[1]).all().item(), 'others equal', torch.equal(gote[[0,2,3]], refe[[0,2,3]]), flag)
        except Exception as e:
            print(precision, 'input', bad, 'ERR', type(e).__name__, str(e)[:150])
    # float64 / int16 tensors through the blocks API
    from diart_amd.blocks import SpeakerSegmentation
    sm = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=8, precision=precision)
    blk = SpeakerSegmentation(sm, gpu)
    x64 = base[:, :, None].cpu().double().numpy()
    out = blk(x64)
    print(precision, 'block float64 numpy ->', type(out).__name__, getattr(out, 'dtype', None), np.abs(np.asarray(out) - ref.numpy()).max())
