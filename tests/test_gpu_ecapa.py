"""Config 3 embedding (speechbrain ECAPA-TDNN behind pyannote's PretrainedSpeakerEmbedding
contract): the HIP path vs the CPU restatement (oracle/ecapa_ref.py), stage by stage and end to
end, including the mask-driven variable lengths and the NaN rows.  Tolerance: fp32 round-off of a
deeper network than the x-vector (features in dB, 3 SE-Res2Net blocks, softmax pooling):
relative L2 <= 2e-4 on every stage, cosine >= 0.99999 on the embeddings."""
import numpy as np
import pytest
import torch

from diart_amd import models as M
from diart_amd.synth import synth_ecapa_state, synth_streams

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.fixture(scope="module")
def oracle():
    from oracle.ecapa_ref import PretrainedSpeakerEmbeddingRef
    return PretrainedSpeakerEmbeddingRef(synth_ecapa_state())


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def hip(gpu, request):
    """Both arithmetic modes against the same gates: "f16x3" runs the wide 1x1 layers (tdnn1, tdnn2,
    mfa: 87 % of the FLOPs) on the split-f16 kernel, "f32" everything on exact-f32 MFMA."""
    return M.HipEcapaEmbedding(synth_ecapa_state(), max_batch=8, precision=request.param).to(gpu)


def test_stages_without_masks(gpu, oracle, hip):
    from oracle.ecapa_ref import fbank, sentence_mean_norm
    S = 24000
    x = torch.from_numpy(synth_streams(3, 1.5, seed0=40))[:, None, :S].contiguous()
    got = hip(x.to(gpu)).cpu()
    with torch.no_grad():
        feats = sentence_mean_norm(fbank(x[:, 0]), torch.ones(3))
        ref, inter = oracle.model(feats, torch.ones(3), return_intermediate=True)
    g_feats, T = hip.peek(S, 0)
    assert T == 1 + S // 160 == feats.shape[1]
    assert rel(g_feats.cpu().view(3, T, 80), feats) < 2e-5
    assert rel(hip.peek(S, 1)[0].cpu().view(3, T, 1024), inter["block0"].transpose(1, 2)) < 2e-5
    assert rel(hip.peek(S, 3)[0].cpu().view(3, T, 3072), inter["mfa"].transpose(1, 2)) < 1e-4
    assert rel(hip.peek(S, 4)[0].cpu().view(3, 6144), inter["pooled"]) < 1e-4
    assert rel(got, ref) < 2e-4
    cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=-1)
    assert cos.min().item() > 0.99999


def test_masks_variable_length_and_nan_rows(gpu, oracle, hip):
    S, Fw = 80000, 293
    x = torch.from_numpy(synth_streams(5, 5.0, seed0=50))[:, None, :S].contiguous()
    g = torch.Generator().manual_seed(3)
    masks = torch.rand(5, Fw, generator=g)          # ~half of the frames kept, scattered
    masks[1, :] = 0.0
    masks[1, 40:90] = 0.8                            # one contiguous 0.85 s turn
    masks[2, :] = 0.0
    masks[2, 10:12] = 1.0                            # 2 frames = 546 samples < 640 -> NaN
    masks[3, :] = 1.0                                # everything kept: the longest row
    masks[4, :] = 0.5                                # exactly 0.5 is NOT > 0.5 -> nothing kept -> NaN
    ref = oracle(x, masks)
    got = hip(x.to(gpu), masks.to(gpu)).cpu().numpy()
    lens = hip.peek(S, 5)[0].cpu().numpy()
    _, want_lens = oracle.select(x, masks)
    assert np.array_equal(lens, want_lens.numpy())
    assert np.isnan(ref[2]).all() and np.isnan(ref[4]).all()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref[:, 0])
    cos = torch.nn.functional.cosine_similarity(torch.from_numpy(got[ok]).double(),
                                                torch.from_numpy(ref[ok]).double(), dim=-1)
    print("ecapa cos", cos.tolist(), "rel", rel(torch.from_numpy(got[ok]), torch.from_numpy(ref[ok])))
    assert cos.min().item() > 0.99999
    assert rel(torch.from_numpy(got[ok]), torch.from_numpy(ref[ok])) < 3e-4
    # the batch quirk is reproduced: dropping the longest row changes the padding and therefore
    # (slightly) the embeddings of the others, exactly as in the reference
    ref2 = oracle(x[:2], masks[:2])
    got2 = hip(x[:2].to(gpu), masks[:2].to(gpu)).cpu().numpy()
    assert rel(torch.from_numpy(got2), torch.from_numpy(ref2)) < 3e-4


def test_all_rows_too_short(gpu, oracle, hip):
    x = torch.from_numpy(synth_streams(2, 5.0, seed0=60))[:, None, :80000].contiguous()
    masks = torch.zeros(2, 293)
    masks[:, 5] = 1.0
    ref = oracle(x, masks)
    got = hip(x.to(gpu), masks.to(gpu)).cpu().numpy()
    assert np.isnan(ref).all() and np.isnan(got).all() and got.shape == (2, 192)


def test_operator_api_with_ecapa(gpu, oracle):
    """EmbeddingModel / SpeakerEmbedding with normalised OSP weights as diart drives them
    (normalize_embedding_weights=True, argdoc.py:18)."""
    from diart_amd.blocks import OverlapAwareSpeakerEmbedding
    from oracle.functional_ref import normalize_embeddings_ref, overlapped_speech_penalty_ref
    model = M.EmbeddingModel.from_state(synth_ecapa_state(), max_batch=6)
    block = OverlapAwareSpeakerEmbedding(model, gamma=3, beta=10, norm=1, normalize_weights=True, device=gpu)
    wav = torch.from_numpy(synth_streams(2, 5.0, seed0=70))[:, :80000, None].contiguous()
    g = torch.Generator().manual_seed(1)
    seg = torch.rand(2, 293, 3, generator=g)
    seg[:, :, 2] *= 0.05
    got = block(wav, seg)
    assert got.shape == (2, 3, 192)
    w = overlapped_speech_penalty_ref(seg)                       # + min-max like the block
    mn, mx = w.min(dim=1, keepdim=True).values, w.max(dim=1, keepdim=True).values
    w = ((w - mn) / (mx - mn)).nan_to_num(1e-8)
    rows = wav.transpose(1, 2).repeat(1, 3, 1).reshape(6, 1, -1)
    ref = torch.from_numpy(oracle(rows, w.permute(0, 2, 1).reshape(6, 293))).view(2, 3, 192)
    ref = normalize_embeddings_ref(ref)
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    ok = ~torch.isnan(ref[..., 0])
    cos = (got[ok].double() * ref[ok].double()).sum(-1)
    assert cos.min().item() > 0.99999


def test_nan_samples_only_count_where_the_mask_keeps_them(gpu, oracle, hip):
    """A NaN / Inf sample the mask KEEPS makes that row's embedding NaN (the reference's fbank of it is NaN); one the
    mask drops never reaches the network.  The row keeps its place in the batch geometry (padding to the longest row),
    so the other rows are bit-identical to the clean batch."""
    S, Fw = 80000, 293
    x = torch.from_numpy(synth_streams(4, 5.0, seed0=70))[:, None, :S].contiguous()
    masks = torch.zeros(4, Fw)
    masks[0, 20:200] = 1.0
    masks[1, :] = 1.0                                # the longest row — and the one with the kept NaN
    masks[2, 100:250] = 1.0
    masks[3, 50:150] = 1.0
    clean = hip(x.to(gpu), masks.to(gpu)).cpu().numpy()
    bad = x.clone()
    bad[1, 0, 33333] = float("nan")                  # kept (row 1 keeps everything)
    bad[2, 0, 100] = float("inf")                    # frame 0 of row 2: dropped by its mask
    bad[3, 0, int(100 * S / Fw)] = float("inf")      # frame 100 of row 3: kept
    ref = oracle(bad, masks)
    got = hip(bad.to(gpu), masks.to(gpu)).cpu().numpy()
    assert np.isnan(ref[1]).all() and np.isnan(ref[3]).all() and np.isfinite(ref[[0, 2]]).all()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(got[[0, 2]], clean[[0, 2]])
    from diart_amd import _lib
    _lib.range_check(gpu.index)                       # a NaN row, not an out-of-range error for the whole call
    lens = hip.peek(S, 5)[0].cpu().numpy()
    assert (lens[[1, 3]] < 0).all() and (lens[[0, 2]] > 0).all()         # -(len + 1) marks a row with non-finite samples
