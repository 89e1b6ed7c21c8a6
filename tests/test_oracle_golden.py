"""CPU: the oracle (oracle/cvlm_oracle.py) is pinned to golden vectors produced by running the
REFERENCE itself (tools/make_golden.py).  These tests are what makes the oracle trustworthy as the
checker of the HIP path."""
import os

import numpy as np
import pytest
import torch

from camouflaged_vlm_amd import spec, synth
from oracle import cvlm_oracle as O


@pytest.fixture(scope="module")
def gold(golden_dir):
    with np.load(os.path.join(golden_dir, "tiny_cascade.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def run(gold):
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    sd = O.to_torch_sd(synth.make_full_state_dict(g, c))
    inp, ci, cm = (torch.from_numpy(t) for t in synth.make_inputs(g, c, 2))
    bank = torch.from_numpy(gold["bank_test"])
    taps = {}
    with torch.no_grad():
        tf = O.clip_text_features(sd, c, gold["eot_test"].tolist())
        m, pred, logits = O.cascade(inp, ci, cm, sd, g, c, tf, bank, taps)
    return dict(g=g, c=c, sd=sd, tf=tf, m=m, pred=pred, logits=logits, taps=taps)


def d(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def test_text_encoder_matches_reference(run, gold):
    assert d(run["tf"], gold["tap_clip_text"]) < 1e-5
    with torch.no_grad():
        tr = O.clip_text_features(run["sd"], run["c"], gold["eot_test"].tolist(), truncate=True)
    assert d(tr, gold["tap_clip_text"]) < 1e-5          # truncation to EOT+1 positions is exact (causal)


def test_encoder_stages_match_reference(run, gold):
    t = run["taps"]
    assert d(t["patch_embed"][:1], gold["tap_patch_embed"]) < 1e-5
    assert d(t["highpass"][:1], gold["highpass"]) < 1e-5
    for i in range(4):
        assert d(t[f"block{i}"][:1], gold[f"tap_block{i}"]) < 2e-5
    assert d(t["features"][:1], gold["tap_features"]) < 2e-5


def test_decoder_stages_match_reference(run, gold):
    t = run["taps"]
    assert d(t["hs"], gold["tap_hs"]) < 2e-5
    assert d(t["src"].flatten(2).transpose(1, 2), gold["tap_src"]) < 2e-5
    assert d(t["upscaled"], gold["tap_upscaled"]) < 2e-5
    assert d(t["low_res_masks"][:1], gold["tap_low_res_masks"]) < 1e-4
    assert d(O.dense_pe(run["sd"], run["g"].grid)[None], gold["dense_pe"]) < 1e-6


def test_cascade_outputs_match_reference(run, gold):
    assert d(run["m"], gold["mask_logits"]) < 1e-4
    assert d(run["logits"], gold["class_logits"]) < 1e-4
    assert d(run["taps"]["pass1_logits"], gold["pass1_logits"]) < 1e-4
    assert run["pred"].tolist() == gold["pred"].tolist()
    assert O.mask_iou(run["m"].numpy(), gold["mask_logits"]) >= 0.9999


def test_batch_equals_per_image(run):
    """The reference only runs B=1; the batched oracle must equal per-image calls."""
    g, c, sd = run["g"], run["c"], run["sd"]
    inp, ci, cm = (torch.from_numpy(t) for t in synth.make_inputs(g, c, 2))
    with torch.no_grad():
        m1, _, l1 = O.cascade(inp[1:], ci[1:], cm[1:], sd, g, c, run["tf"], torch.from_numpy(
            np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_cascade.npz"))["bank_test"]))
    assert d(m1, run["m"][1:]) < 1e-5 and d(l1, run["logits"][1:]) < 1e-5


def test_tokens_fixture(golden_dir):
    with np.load(os.path.join(golden_dir, "ovcamo_constants.npz")) as z:
        tk = z["tokens_test"]
        assert tk.shape == (61, 77) and z["tokens_train"].shape == (14, 77)
        # "a photo of a owlfly larva." (measured from the reference tokenizer)
        assert tk[0, :11].tolist() == [49406, 320, 1125, 539, 320, 34332, 3228, 1592, 1892, 269, 49407]
        assert z["bank_test"].shape == (61, 768) and abs(float(np.linalg.norm(z["bank_test"][0])) - 1) < 1e-4


def test_plain_sam_entry_matches_reference(golden_dir):
    """N4: registry entry `sam` (vanilla MaskDecoder, no prompts) -- oracle vs the reference's `SAM.infer`."""
    with np.load(os.path.join(golden_dir, "tiny_sam_plain.npz")) as z:
        gp = {k: z[k] for k in z.files}
    g = spec.TINY_SAM
    sd = O.to_torch_sd(synth.make_state_dict(spec.sam_plain_entries(g), 0))
    inp, _, _ = synth.make_inputs(g, spec.TINY_CLIP, 2)
    with torch.no_grad():
        m = O.sam_plain_infer(torch.from_numpy(inp), sd, g)
    assert m.shape == gp["mask_logits"].shape
    assert d(m, gp["mask_logits"]) < 2e-5
    assert O.mask_iou(m.numpy(), gp["mask_logits"]) >= 0.9999


def test_oracle_matches_reference_on_outlier_weights(golden_dir):
    """The oracle against the reference run on synth.apply_outliers weights (massive residual channels, hot MLP units)."""
    from camouflaged_vlm_amd import spec, synth
    from oracle import cvlm_oracle as O
    with np.load(os.path.join(golden_dir, "tiny_outliers.npz")) as z:
        go = {k: z[k] for k in z.files}
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    sd = O.to_torch_sd(synth.apply_outliers(synth.make_full_state_dict(g, c)))
    inp, ci, cm = (torch.from_numpy(t) for t in synth.make_inputs(g, c, batch=2))
    with torch.no_grad():
        tf = O.clip_text_features(sd, c, go["eot_test"].tolist())
        m, p, l = O.cascade(inp, ci, cm, sd, g, c, tf, torch.from_numpy(go["bank_test"]))
    assert float((m - torch.from_numpy(go["mask_logits"])).abs().max()) < 1e-4
    assert float((l - torch.from_numpy(go["class_logits"])).abs().max()) < 1e-4
    assert p.tolist() == go["pred"].tolist()


def test_digest_checker_accepts_the_reference_and_rejects_deviations(golden_dir):
    """camouflaged_vlm_amd.digest (the gate bench.py and the GPU tests share) on synthetic outputs built FROM the digest: a mask
    with the reference's sign bits and sampled logits passes for every image; a flipped region, a logit off by 2e-3, a wrong
    prediction each fail; image ids the digest does not hold are skipped, not passed silently as 'checked'."""
    import torch
    from camouflaged_vlm_amd import digest
    dg = digest.load(os.path.join(golden_dir, "demo_digest.npz"))
    n = digest.n_images(dg)
    assert n == 16 and dg["mask_bits"].shape == (16, 131072) and dg["feat_samples"].shape == (16, 16384)
    ids = [0, 5, 15]
    S = 1024

    def outputs():
        masks = []
        for i in ids:
            m = np.where(np.unpackbits(dg["mask_bits"][i])[:S * S].astype(bool), 1.0, -1.0).astype(np.float32)
            m[dg["sample_idx"]] = dg["mask_samples"][i]
            m[dg["dense_idx"]] = dg["dense_samples"][i]              # round 6: 65536 common positions + the image's 4096 smallest |logit|
            m[dg["near_idx"][i]] = dg["near_samples"][i]
            masks.append(m.reshape(1, S, S))
        return (torch.from_numpy(np.stack(masks)), torch.from_numpy(dg["pred"][ids].copy()),
                torch.from_numpy(dg["class_logits"][ids].copy()))

    m, p, l = outputs()
    r = digest.check_cascade(m, p, l, dg, ids)
    assert r["ok"] and r["checked_images"] == ids and r["min_iou"] == 1.0 and r["max_abs_mask_err"] == 0.0
    assert r["mask_positions_per_image"] == 4096 + 65536 + 4096 and set(r["max_abs_mask_err_by_set"]) == {"sparse", "dense", "near"}
    m3 = m.clone()
    m3.view(3, -1)[0, int(dg["near_idx"][0][7])] += 2e-3              # one pixel next to the decision boundary off by 2e-3: only the near set sees it
    r3 = digest.check_cascade(m3, p, l, dg, ids)
    assert not r3["ok"] and r3["max_abs_mask_err_by_set"]["near"] > 1e-3
    m2 = m.clone()
    m2[1, 0, :64, :64] *= -1.0                                       # flip a 64 x 64 patch of image 5
    assert not digest.check_cascade(m2, p, l, dg, ids)["ok"]
    l2 = l.clone()
    l2[2, 7] += 2e-3
    assert not digest.check_cascade(m, p, l2, dg, ids)["ok"]
    p2 = p.clone()
    p2[0] = (p2[0] + 1) % 61
    assert not digest.check_cascade(m, p2, l, dg, ids)["ok"]
    r = digest.check_cascade(m, p, l, dg, [0, 16, 99])               # only image 0 exists in the digest
    assert r["checked_images"] == [0]
    assert digest.check_cascade(m[:1], p[:1], l[:1], dg, [40])["checked_images"] == []
    # encoder features: NHWC tokens rebuilt from nothing but the samples cannot pass the channel-mean gate
    f = torch.zeros(64 * 64, 256)
    assert not digest.check_demo_features(f, 64, dg, [0])["ok"]
