"""One-class scoring pipeline on the GPU (oc_classifier mirror) vs the CPU oracle chain on synthetic 16-bit wav files:
reference embedding = mean bona-fide embedding, threshold = largest distance, scores.txt lines "{dist}, {0|1} \\n"."""
import os
import wave

import numpy as np
import pytest
import torch

from conftest import synth_wave

pytestmark = pytest.mark.gpu


def _write_wav(path, n, seed):
    x = (np.clip(synth_wave(seed, n), -1, 1) * 32767).astype(np.int16)
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())


def _write_flac(path, n, seed):
    """The evaluation side of the corpus as real FLAC (the ASVspoof corpora are .flac): written by the test-only encoder, read back by
    the library's host decoder inside ASVDataset.  Returns the samples the oracle should see."""
    try:
        from tests import flac_writer as fw
    except ImportError:
        import flac_writer as fw
    x = (np.clip(synth_wave(seed, n), -1, 1) * 32767).astype(np.int16)
    with open(path, "wb") as f:
        f.write(fw.encode(x.astype(np.int64)[:, None], blocksize=4096, kinds=("lpc", "fixed2"), porder=3, lpc=([1520, -640, 90], 12, 10)))
    return x.astype(np.float32) / 32768.0


def test_reference_embedding_threshold_and_scores_match_oracle(tmp_path, monkeypatch):
    from oracle import aasist_ref, losses_ref, xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.data_utils_SSL import load_audio
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.oc_classifier import ASVDataset, create_reference_embedding2, score_eval_set_1c2
    from torch.utils.data import DataLoader
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "audio"; d.mkdir()
    tr_lines, ev_lines = [], []
    for i in range(5):
        _write_wav(str(d / f"T{i}.wav"), 9000 + 400 * i, 10 + i)
        tr_lines.append(f"LA_{i} T{i} - - {'bonafide' if i != 2 else 'spoof'}")
    eval_pcm = {}
    for i in range(4):
        eval_pcm[i] = _write_flac(str(d / f"E{i}.flac"), 8000 + 700 * i, 50 + i)
        ev_lines.append(f"E{i}")
    (tmp_path / "train.txt").write_text("\n".join(tr_lines) + "\n")
    (tmp_path / "eval.txt").write_text("\n".join(ev_lines) + "\n")
    kw = dict(dim=1024, ffn=512, heads=16, layers=1)
    rcfg, cfg = xlsr_ref.XlsrConfig(**kw), xlsr.XlsrConfig(**kw)
    px = fill_like(xlsr_ref.param_shapes(rcfg), seed=3)
    pb = fill_like(aasist_ref.param_shapes(), seed=0)
    model = AModel(None, "cuda", ssl_cfg=cfg, ssl_dtype=torch.float32, ssl_state_dict=px, backend_state_dict=pb)
    tr = DataLoader(ASVDataset(str(tmp_path / "train.txt"), str(d)), batch_size=1, shuffle=False)
    ev = DataLoader(ASVDataset(str(tmp_path / "eval.txt"), str(d), eval=True), batch_size=1, shuffle=False)
    ref_emb, thr = create_reference_embedding2(model, tr, "cuda")
    score_eval_set_1c2(model, ev, "cuda", ref_emb, thr)

    def oracle_emb(path):
        x = eval_pcm[int(os.path.basename(path)[1])] if path.endswith(".flac") else load_audio(path)[0]
        with torch.no_grad():
            f = xlsr_ref.extract_feat(torch.tensor(x)[None], px, rcfg)
            return aasist_ref.backend_forward(f, pb, train=False)[0]
    embs = torch.stack([oracle_emb(str(d / f"T{i}.wav")) for i in (0, 1, 3, 4)])          # bona-fide rows only
    o_ref, o_thr, o_dist = losses_ref.reference_embedding_and_threshold(embs)
    torch.testing.assert_close(ref_emb.cpu(), o_ref, rtol=1e-3, atol=1e-3)
    assert abs(float(thr) - float(o_thr)) < 2e-3
    assert os.path.exists("reference_embedding.pt") and os.path.exists("threshold.pt")
    got = [float(l.split(",")[0]) for l in open("distances.txt")]
    np.testing.assert_allclose(got, o_dist.numpy(), rtol=1e-3, atol=2e-3)
    lines = open("scores.txt").read().splitlines()
    assert len(lines) == 4
    for i, line in enumerate(lines):
        dist, flag = line.split(",")
        od = float(losses_ref.pairwise_l2(o_ref, oracle_emb(str(d / f"E{i}.flac"))))
        assert abs(float(dist) - od) < 2e-3 and line.endswith(" ")
        assert int(flag) == int(float(dist) > float(thr))


def test_extractor_encoder_scoring_variants_match_oracle(tmp_path, monkeypatch):
    """The two-model scoring functions (oc_classifier.py:113-157 create_reference_embedding, :206-241 score_eval_set_1c1, :268-291
    score_eval_set_2c1): XLS-R extractor + SE-ResNet34 encoder, f32 path, against xlsr_ref -> senet_ref on the same files."""
    from oracle import losses_ref, senet_ref, xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.data_utils_SSL import load_audio
    from occm_amd.models import xlsr
    from occm_amd.models.senet import se_resnet34
    from occm_amd.oc_classifier import ASVDataset, create_reference_embedding, score_eval_set_1c1, score_eval_set_2c1
    from torch.utils.data import DataLoader
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "audio"; d.mkdir()
    tr_lines, ev_lines = [], []
    for i in range(4):
        _write_wav(str(d / f"T{i}.wav"), 9000 + 500 * i, 20 + i)
        tr_lines.append(f"LA_{i} T{i} - - {'bonafide' if i != 1 else 'spoof'}")
    for i in range(3):
        _write_wav(str(d / f"E{i}.wav"), 8200 + 900 * i, 70 + i)
        ev_lines.append(f"E{i}")
    (tmp_path / "train.txt").write_text("\n".join(tr_lines) + "\n")
    (tmp_path / "eval.txt").write_text("\n".join(ev_lines) + "\n")
    kw = dict(dim=1024, ffn=512, heads=16, layers=1)
    rcfg, cfg = xlsr_ref.XlsrConfig(**kw), xlsr.XlsrConfig(**kw)
    px = fill_like(xlsr_ref.param_shapes(rcfg), seed=3)
    ps = fill_like(senet_ref.param_shapes(), seed=1)
    extractor = xlsr.SSLModel("cuda", state_dict=px, cfg=cfg, dtype=torch.float32)
    encoder = se_resnet34(state_dict=ps, device="cuda")
    tr = DataLoader(ASVDataset(str(tmp_path / "train.txt"), str(d)), batch_size=1, shuffle=False)
    ev = DataLoader(ASVDataset(str(tmp_path / "eval.txt"), str(d), eval=True), batch_size=1, shuffle=False)
    ref_emb, thr = create_reference_embedding(extractor, encoder, tr, "cuda")
    assert tuple(ref_emb.shape) == (1, 128) and os.path.exists("reference_embedding.pt") and os.path.exists("threshold.pt") and not os.path.exists("distances.txt")
    score_eval_set_1c1(extractor, encoder, ev, "cuda", ref_emb, thr, path="scores_1c1.txt")
    score_eval_set_2c1(extractor, encoder, ev, "cuda", path="scores_2c1.txt")

    def oracle_out(path):
        with torch.no_grad():
            f = xlsr_ref.extract_feat(torch.tensor(load_audio(path)[0])[None], px, rcfg)
            return senet_ref.senet34_forward(f.unsqueeze(1), ps, train=False)
    embs = torch.stack([oracle_out(str(d / f"T{i}.wav"))[0] for i in (0, 2, 3)])
    o_ref, o_thr, _ = losses_ref.reference_embedding_and_threshold(embs)
    torch.testing.assert_close(ref_emb.cpu(), o_ref, rtol=1e-3, atol=1e-3)
    assert abs(float(thr) - float(o_thr)) < 2e-3
    l1, l2 = open("scores_1c1.txt").read().splitlines(), open("scores_2c1.txt").read().splitlines()
    assert len(l1) == 3 and len(l2) == 3
    for i in range(3):
        emb, out = oracle_out(str(d / f"E{i}.wav"))
        dist, flag = l1[i].split(",")
        assert abs(float(dist) - float(losses_ref.pairwise_l2(o_ref, emb))) < 2e-3 and l1[i].endswith(" ")
        assert int(flag) == int(float(dist) > float(thr))
        assert abs(float(l2[i]) - float(out[0][0])) < 2e-3


def test_bucketed_batch_scoring_equals_one_at_a_time(tmp_path, monkeypatch):
    """--batch_size 3: utterances are grouped by frame count and cropped to the samples those frames depend on; scores.txt, the reference
    embedding and the threshold must equal the reference's one-utterance loop (f32 front-end: to 2e-4; the two differ only in the GEMM
    kernel picked for a different row count), in dataset order, with ragged lengths, a bucket that never fills and repeated frame counts."""
    from oracle import aasist_ref, xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.oc_classifier import ASVDataset, canonical_len, create_reference_embedding2, embed_dataset, n_frames, score_eval_set_1c2
    from torch.utils.data import DataLoader
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "audio"; d.mkdir()
    lens = [9000, 8720, 9039, 12000, 8800, 12100, 8900, 15000, 8999, 12239, 8721]           # frame counts 27 x7 (two full buckets + 1), 37 x3, 46 x1
    assert sorted(set(n_frames(L) for L in lens)) == [27, 37, 46] and canonical_len(27) == 8720
    lines = []
    for i, L in enumerate(lens):
        _write_wav(str(d / f"U{i}.wav"), L, 100 + i)
        lines.append(f"LA_{i} U{i} - - bonafide")
    (tmp_path / "train.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "eval.txt").write_text("\n".join(f"U{i}" for i in range(len(lens))) + "\n")
    kw = dict(dim=1024, ffn=512, heads=16, layers=1)
    rcfg, cfg = xlsr_ref.XlsrConfig(**kw), xlsr.XlsrConfig(**kw)
    model = AModel(None, "cuda", ssl_cfg=cfg, ssl_dtype=torch.float32, ssl_state_dict=fill_like(xlsr_ref.param_shapes(rcfg), seed=3),
                   backend_state_dict=fill_like(aasist_ref.param_shapes(), seed=0))
    tr = DataLoader(ASVDataset(str(tmp_path / "train.txt"), str(d)), batch_size=1, shuffle=False)
    ev = DataLoader(ASVDataset(str(tmp_path / "eval.txt"), str(d), eval=True), batch_size=1, shuffle=False)
    e1, l1 = embed_dataset(model, ev, "cuda", 1)
    e3, l3 = embed_dataset(model, ev, "cuda", 3)
    torch.testing.assert_close(e3, e1, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(l3, l1, rtol=2e-4, atol=2e-4)
    out = {}
    for bs in (1, 3):
        ref_emb, thr = create_reference_embedding2(model, tr, "cuda", cache=False, batch_size=bs)
        score_eval_set_1c2(model, ev, "cuda", ref_emb, thr, path="scores_%d.txt" % bs, batch_size=bs)
        out[bs] = (ref_emb, float(thr), [l.split(",") for l in open("scores_%d.txt" % bs).read().splitlines()])
    torch.testing.assert_close(out[3][0], out[1][0], rtol=2e-4, atol=2e-4)
    assert abs(out[3][1] - out[1][1]) < 2e-4 and len(out[3][2]) == len(lens)
    for (d3, f3), (d1, f1) in zip(out[3][2], out[1][2]):
        assert abs(float(d3) - float(d1)) < 2e-4
    assert len(model.ssl_model.model._ws) <= 12 and len(model.backend._ws) <= 12


@pytest.mark.parametrize("extra", [[], ["--backend", "senet"], ["--finetuned"], ["--rawboost_algo", "5", "--rawboost_on_gpu"]])
def test_oc_training_entry_point_runs_and_saves_checkpoint(tmp_path, monkeypatch, extra):
    """python -m occm_amd.oc_training on a tiny synthetic corpus (PFDataset groups of 12 from 16-bit wav files): one epoch through the
    real entry point, checkpoint written under the reference's file name with the reference's key names."""
    import random
    from occm_amd import oc_training
    from occm_amd.models import xlsr
    from occm_amd.oc_training import VOCODERS
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(xlsr.XlsrConfig, "xlsr_300m", staticmethod(lambda: xlsr.XlsrConfig(dim=1024, ffn=256, heads=16, layers=1)))
    d, v = tmp_path / "wav", tmp_path / "voc"
    d.mkdir(); v.mkdir()
    lines = []
    for i in range(7):
        # one utterance of 5.7 s: PFDataset zero-pads the group to it (oc_training.py:244-249), i.e. 284 frames -- longer than one key
        # block of the attention backward, which the reference's variable-length groups need when XLS-R is fine-tuned
        lines.append(f"LA_00{i} B{i} - - bonafide"); _write_wav(str(d / f"B{i}.wav"), 91000 if i == 2 else 16000 + 300 * i, i)
        for k, name in enumerate(VOCODERS):
            _write_wav(str(v / f"{name}_B{i}.wav"), 16000 + 50 * k, 100 + i * 5 + k)
    for i in range(3):
        lines.append(f"LA_01{i} S{i} - A0{i} spoof"); _write_wav(str(d / f"S{i}.wav"), 15000 + 11 * i, 50 + i)
    (tmp_path / "prot.txt").write_text("\n".join(lines) + "\n")
    # a fairseq-shaped checkpoint file: {"model": path tensors + the off-path tensors a real xlsr2_300m.pt also holds, "cfg": {...}}
    small = xlsr.XlsrConfig.xlsr_300m()
    ck_model = dict(xlsr.synthetic_params(small, 0))
    g = torch.Generator().manual_seed(9)
    extras = {"mask_emb": torch.rand(1024, generator=g), "quantizer.vars": torch.rand(1, 640, 384, generator=g),
              "quantizer.weight_proj.weight": torch.rand(640, 512, generator=g), "quantizer.weight_proj.bias": torch.rand(640, generator=g),
              "project_q.weight": torch.rand(768, 768, generator=g), "project_q.bias": torch.rand(768, generator=g),
              "final_proj.weight": torch.rand(768, 1024, generator=g), "final_proj.bias": torch.rand(768, generator=g)}
    ck_model.update(extras)
    torch.save({"model": ck_model, "cfg": {"model": {"dropout": 0.0, "attention_dropout": 0.0, "activation_dropout": 0.0, "encoder_layerdrop": 0.0,
                                                      "dropout_input": 0.0, "feature_grad_mult": 1.0}}, "args": None}, str(tmp_path / "xlsr.pt"))
    random.seed(0); torch.manual_seed(0)
    oc_training.main(["--train_protocol_file", str(tmp_path / "prot.txt"), "--train_dataset_dir", str(d), "--vocoded_dir", str(v), "--epochs", "1",
                      "--lr", "1e-4", "--ssl_checkpoint", str(tmp_path / "xlsr.pt")] + extra)
    sd = torch.load(str(tmp_path / "aasist_vocoded_0.pt"), map_location="cpu")
    assert all(torch.isfinite(t.float()).all() for t in sd.values() if torch.is_tensor(t))
    pre = "frontend.model." if "--backend" in extra else "ssl_model.model."
    # the file holds EVERY tensor of the fairseq checkpoint under the reference's prefix (what its strict load_state_dict needs,
    # oc_classifier.py:340), the off-path ones bit-identical, the path's ones with their shapes
    assert {k[len(pre):] for k in sd if k.startswith(pre)} == set(ck_model)
    for k, t in extras.items():
        assert torch.equal(sd[pre + k], t)
    for k, t in ck_model.items():
        assert tuple(sd[pre + k].shape) == tuple(t.shape), k
    trained = sum(int(not torch.equal(sd[pre + k].float(), ck_model[k])) for k in ck_model if k not in extras)
    assert (trained > 50) if "--finetuned" in extra else (trained == 0)
    if "--backend" in extra:
        assert any(k.endswith("layer1.0.conv1.weight") for k in sd)
    else:
        assert "LL.weight" in sd and "out_layer.weight" in sd
        # and it loads back, strictly, into a fresh model (the scorer's path); a file lacking one path tensor is rejected
        from occm_amd.models.sslassist import AModel
        from occm_amd._lib import OccError
        ssl = {k[len(pre):]: t for k, t in sd.items() if k.startswith(pre)}
        m2 = AModel(None, "cuda", ssl_state_dict=ssl)
        m2.load_state_dict(sd, strict=True)
        assert set(m2.state_dict()) == set(sd)
        broken = {k: t for k, t in sd.items() if k != pre + "encoder.layers.0.fc1.weight"}
        with pytest.raises(OccError):
            m2.load_state_dict(broken, strict=True)
        wrong = dict(sd); wrong[pre + "encoder.layers.0.fc1.bias"] = torch.zeros(7)
        with pytest.raises(OccError):
            m2.load_state_dict(wrong, strict=True)


def test_scoring_300m_24_layers_f32_and_bf16_paths_vs_oracle_chain(tmp_path, monkeypatch):
    """The deliverable of oc_classifier.py:159-202, 243-265 at the real model size: XLS-R-300M (24 layers) + AASIST, 16 bona-fide
    utterances for the reference embedding and 48 labelled trial utterances of mixed length (1 - 4 s), scored one utterance per
    forward (the reference's loop) through create_reference_embedding2 / score_eval_set_1c2, with the f32-MFMA path (the entry point's
    default, --ssl_dtype f32) and with the bf16 path (--ssl_dtype bf16), against the CPU oracle chain on the same files.  Prints and
    bounds: max |d emb|, max |d distance|, flipped threshold decisions and d EER (occm_amd.evaluate_metrics.compute_eer, spoof = target as
    evaluate.py:143-145).  north_star: embeddings / scores within 1e-3, EER within +-0.2 (percent)."""
    import json
    from oracle import aasist_ref, losses_ref, xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.data_utils_SSL import load_audio
    from occm_amd.evaluate_metrics import compute_eer
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.oc_classifier import ASVDataset, create_reference_embedding2, score_eval_set_1c2
    from torch.utils.data import DataLoader
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "audio"; d.mkdir()
    rs = np.random.RandomState(7)
    lens = [int(v) for v in rs.randint(16000, 64001, size=64)]
    lens[0], lens[1], lens[16], lens[17] = 64000, 16000, 64000, 16000
    n_ref, n_ev = 16, 48

    def write(path, n, seed, spoof):
        # "spoofed" trials are a different signal family (an AM tone under the noise) so that distances spread on both sides of the threshold
        x = synth_wave(seed, n)
        if spoof:
            t = np.arange(n, dtype=np.float32) / 16000.0
            x = 0.6 * x + 0.08 * np.sin(2 * np.pi * (300 + 40 * (seed % 7)) * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))
        pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
        with wave.open(path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())

    tr_lines, ev_lines, is_spoof = [], [], []
    for i in range(n_ref):
        write(str(d / f"T{i}.wav"), lens[i], 1000 + i, False)
        tr_lines.append(f"LA_{i} T{i} - - bonafide")
    for i in range(n_ev):
        sp = i % 2 == 1
        write(str(d / f"E{i}.wav"), lens[n_ref + i], 2000 + i, sp)
        ev_lines.append(f"E{i}"); is_spoof.append(sp)
    (tmp_path / "train.txt").write_text("\n".join(tr_lines) + "\n")
    (tmp_path / "eval.txt").write_text("\n".join(ev_lines) + "\n")
    is_spoof = np.array(is_spoof)

    rcfg, cfg = xlsr_ref.XlsrConfig.xlsr_300m(), xlsr.XlsrConfig.xlsr_300m()
    px = fill_like(xlsr_ref.param_shapes(rcfg), seed=0)
    pb = fill_like(aasist_ref.param_shapes(), seed=0)

    def oracle_emb(path):
        x = load_audio(path)[0]
        with torch.no_grad():
            f = xlsr_ref.extract_feat(torch.tensor(x)[None], px, rcfg)
            return aasist_ref.backend_forward(f, pb, train=False)[0]
    o_tr = torch.stack([oracle_emb(str(d / f"T{i}.wav")) for i in range(n_ref)])
    o_ev = torch.stack([oracle_emb(str(d / f"E{i}.wav")) for i in range(n_ev)])
    o_ref, o_thr, _ = losses_ref.reference_embedding_and_threshold(o_tr)
    o_dist = np.array([float(losses_ref.pairwise_l2(o_ref, e)) for e in o_ev])
    # a threshold that separates the trial set (the bona-fide maximum of 16 utterances may sit above or below every trial): the median trial
    # distance, so flipped decisions are possible and counted; the bona-fide-maximum threshold itself is compared as a number
    o_cut = float(np.median(o_dist))
    # Labels for the EER: random weights do not separate the two signal families (oracle EER 50 %), so the trial labels are derived from
    # the ORACLE's ranking -- spoof = upper half of its distances, with every 5th trial by rank swapped across the cut so the classes
    # overlap.  d EER then measures how far a path's distance errors re-rank the trials.
    rank = np.argsort(np.argsort(o_dist))
    is_spoof = (rank >= n_ev // 2) ^ (rank % 5 == 2)
    o_eer = compute_eer(o_dist[is_spoof], o_dist[~is_spoof])[0] * 100.0
    report = {"n_reference": n_ref, "n_trials": n_ev, "oracle": {"threshold": float(o_thr), "eer_percent": o_eer, "dist_min": float(o_dist.min()), "dist_max": float(o_dist.max()),
                                                                   "emb_abs_max": float(o_ev.abs().max())}}
    from occm_amd.oc_classifier import embed_dataset
    # third arm: bf16 front-end with the exact-f32 back-end, to tell the two sources of the bf16 path's error apart
    # fourth arm (round 4): f32 storage with split-operand GEMMs (three bf16 MFMAs per product block, OCC_F32X3) -- must land in the f32 arm's class
    for tag, dt, bc in (("f32", torch.float32, None), ("f32x3", torch.float32, None), ("bf16", torch.bfloat16, None), ("bf16_frontend_f32_backend", torch.bfloat16, "f32")):
        model = AModel(None, "cuda", ssl_cfg=cfg, ssl_dtype=dt, ssl_state_dict=px, backend_state_dict=pb, backend_compute=bc,
                       ssl_f32_gemm="x3" if tag == "f32x3" else "exact")
        tr = DataLoader(ASVDataset(str(tmp_path / "train.txt"), str(d)), batch_size=1, shuffle=False)
        ev = DataLoader(ASVDataset(str(tmp_path / "eval.txt"), str(d), eval=True), batch_size=1, shuffle=False)
        ref_emb, thr = create_reference_embedding2(model, tr, "cuda", cache=False)
        score_eval_set_1c2(model, ev, "cuda", ref_emb, o_cut, path="scores_%s.txt" % tag)
        rows = [l.split(",") for l in open("scores_%s.txt" % tag).read().splitlines()]
        dist = np.array([float(a) for a, _ in rows]); flag = np.array([int(b) for _, b in rows])
        emb_ev, _ = embed_dataset(model, ev, "cuda", 1)
        eer = compute_eer(dist[is_spoof], dist[~is_spoof])[0] * 100.0
        r = {"max_abs_d_emb": float((emb_ev.cpu() - o_ev.reshape(n_ev, -1)).abs().max()), "max_abs_d_ref_emb": float((ref_emb.cpu() - o_ref).abs().max()),
             "max_abs_d_distance": float(np.abs(dist - o_dist).max()), "d_threshold": abs(float(thr) - float(o_thr)),
             "flipped_decisions": int((flag != (o_dist > o_cut).astype(int)).sum()), "eer_percent": eer, "d_eer_percent": abs(eer - o_eer),
             "rel_d_distance": float((np.abs(dist - o_dist) / o_dist).max())}
        report[tag] = r
        del model
        torch.cuda.empty_cache()
    print("scoring parity 300M x 24 layers: " + json.dumps(report))
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "scoring_parity_300m.json"), "w") as f:
            json.dump(report, f, indent=1)
    f32, b16 = report["f32"], report["bf16"]
    # the default path of the entry point: north_star's 1e-3 on embeddings and scores, EER within +-0.2
    assert f32["max_abs_d_emb"] < 1e-3 and f32["max_abs_d_distance"] < 1e-3 and f32["d_threshold"] < 1e-3, f32
    assert f32["d_eer_percent"] <= 0.2, f32
    # a decision can only flip for a trial whose oracle distance lies within the distance error of the cut
    near = int((np.abs(o_dist - o_cut) < 1e-3).sum())
    assert f32["flipped_decisions"] <= near, (f32, near)
    x3 = report["f32x3"]                          # the same bars for the split-operand arithmetic (oc_classifier --ssl_dtype f32x3)
    assert x3["max_abs_d_emb"] < 1e-3 and x3["max_abs_d_distance"] < 1e-3 and x3["d_threshold"] < 1e-3 and x3["d_eer_percent"] <= 0.2, x3
    assert x3["flipped_decisions"] <= near, (x3, near)
    # bf16 path (--ssl_dtype bf16): the measured effect through a RANDOM-weight AASIST (embedding entries up to ~11, distances 3 - 10) is
    # max |d emb| 2.5, max |d distance| 1.0, one flipped decision of 48, EER moved by one trial (DESIGN.md section 5 quotes this run);
    # bounded at about twice that.  It does not meet north_star's 1e-3: that is why f32 is the scoring default.
    assert b16["max_abs_d_emb"] < 5.0 and b16["max_abs_d_distance"] < 2.0, b16
    assert b16["flipped_decisions"] <= 4 and b16["d_eer_percent"] <= 8.4, b16


@pytest.mark.parametrize("T,H,hd,dt", [(50, 4, 64, torch.float32), (199, 16, 64, torch.float32), (343, 4, 64, torch.float32), (700, 2, 64, torch.float32),
                                       (700, 2, 80, torch.float32), (1030, 1, 128, torch.float32), (420, 4, 64, torch.bfloat16),
                                       # block edges of the f32 matrix-core kernel (64 queries per workgroup, keys in blocks of 64, 16-key tiles)
                                       (8, 1, 64, torch.float32), (64, 2, 64, torch.float32), (65, 2, 80, torch.float32), (129, 3, 64, torch.float32)])
def test_f32_attention_any_length_and_key_masks(T, H, hd, dt):
    """occ_attention on the f32 scoring path used to stop at ~300 frames (K and V of a head in LDS); utterances of the ASVspoof eval lists
    run to 13 s (650 frames).  The streaming kernels (keys in blocks of 64, online softmax: on the f32 matrix cores for f32 storage and head
    dims 64 / 80 -- csrc/attention_f32.hip --, the VALU form otherwise) take any length, and occ_attention_varlen masks the pad keys of a
    zero-padded batch: rows [0, len_b) of every utterance against torch's softmax attention of the un-padded rows."""
    from occm_amd import ops
    B, D = 3, H * hd
    g = torch.Generator().manual_seed(T + hd)
    qkv = torch.randn(B * T, 3 * D, generator=g).to(dt).cuda()
    lens = [T, max(1, T // 3), T - 7]

    def ref(b, n):
        x = qkv.view(B, T, 3, H, hd)[b, :n].double()
        q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)           # [H, n, hd]
        p = torch.softmax(q @ k.transpose(1, 2) * hd ** -0.5, dim=-1)
        return (p @ v).transpose(0, 1).reshape(n, D)

    tol = dict(rtol=1e-4, atol=2e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    full = ops.attention(qkv, B, T, H, hd, hd ** -0.5).view(B, T, D)
    for b in range(B):
        torch.testing.assert_close(full[b].double(), ref(b, T), **tol)
    kv = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = ops.attention_varlen(qkv, B, T, H, hd, hd ** -0.5, kv).view(B, T, D)
    assert bool(torch.isfinite(out.float()).all())                                                     # pad rows: not meaningful, but never NaN
    for b, n in enumerate(lens):
        torch.testing.assert_close(out[b, :n].double(), ref(b, n), **tol)


def test_masked_batches_of_unequal_utterances_equal_one_at_a_time(tmp_path, monkeypatch):
    """oc_classifier --batch_size N (length-bucketed batches WITH masks): 14 utterances of 13 distinct frame counts between 1.1 s
    and 7.2 s (the longest past the whole-head limit of the f32 attention kernel), two layers, f32.  Every utterance's embedding and logits in
    masked batches of 4 must equal the reference's one-utterance loop; the un-masked padded forward must NOT (the mask is doing something)."""
    from oracle import aasist_ref, xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.oc_classifier import ASVDataset, embed_dataset, n_frames
    from torch.utils.data import DataLoader
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "audio"; d.mkdir()
    lens = [17600, 18000, 19000, 20100, 21000, 22222, 23000, 24000, 64000, 64000, 66000, 67000, 112000, 115000]
    frames = [n_frames(L) for L in lens]
    assert len(set(frames)) >= 12 and max(frames) > 340
    for i, L in enumerate(lens):
        _write_wav(str(d / f"U{i}.wav"), L, 300 + i)
    (tmp_path / "eval.txt").write_text("\n".join(f"U{i}" for i in range(len(lens))) + "\n")
    kw = dict(dim=1024, ffn=512, heads=16, layers=2)
    rcfg, cfg = xlsr_ref.XlsrConfig(**kw), xlsr.XlsrConfig(**kw)
    model = AModel(None, "cuda", ssl_cfg=cfg, ssl_dtype=torch.float32, ssl_state_dict=fill_like(xlsr_ref.param_shapes(rcfg), seed=3),
                   backend_state_dict=fill_like(aasist_ref.param_shapes(), seed=0))
    ev = DataLoader(ASVDataset(str(tmp_path / "eval.txt"), str(d), eval=True), batch_size=1, shuffle=False)
    e1, l1 = embed_dataset(model, ev, "cuda", 1)
    calls = []
    fwd = model.forward
    model.forward = lambda x, **kw_: (calls.append((tuple(x.shape), kw_.get("lengths"))), fwd(x, **kw_))[1]
    e4, l4 = embed_dataset(model, ev, "cuda", 4)
    model.forward = fwd
    assert len(calls) <= 6 and any(c[1] is not None and len(set(c[1])) > 1 for c in calls), calls     # 14 utterances in at most 6 forwards, some ragged
    torch.testing.assert_close(e4, e1, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(l4, l1, rtol=2e-4, atol=2e-4)
    # the same padded batch WITHOUT the mask is a different function of the shorter utterance
    model.eval()
    from occm_amd.data_utils_SSL import load_audio
    a, b = [torch.tensor(load_audio(str(d / f"U{i}.wav"))[0]) for i in (0, 7)]
    x = torch.zeros(2, b.numel()); x[0, : a.numel()] = a; x[1] = b
    with torch.no_grad():
        em, _ = model(x.cuda(), lengths=[a.numel(), b.numel()])
        feats = model.ssl_model.extract_feat(x.cuda())[:1, : n_frames(a.numel())].contiguous()
        eu, _ = model.backend.forward(feats, train=False)
    torch.testing.assert_close(em[0], e1[0], rtol=2e-4, atol=2e-4)
    assert float((eu[0] - e1[0]).abs().max()) > 1e-2


def test_two_model_scoring_form_in_masked_batches_equals_one_at_a_time():
    """The (extractor, encoder) pair of oc_classifier.py:139-144 -- SSLModel features -> unsqueeze(1) -> se_resnet34 -- through
    ExtractorEncoder with lengths: the front-end runs the zero-padded batch with key masks, the encoder once per distinct frame count."""
    from oracle import xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    from occm_amd.models.senet import se_resnet34
    from occm_amd.oc_classifier import ExtractorEncoder
    kw = dict(dim=1024, ffn=512, heads=16, layers=1)
    ssl = xlsr.SSLModel("cuda", state_dict=fill_like(xlsr_ref.param_shapes(xlsr_ref.XlsrConfig(**kw)), seed=3), cfg=xlsr.XlsrConfig(**kw), dtype=torch.float32)
    enc = se_resnet34(compute="f32")
    pair = ExtractorEncoder(ssl, enc).eval()
    lens = [24000, 17777, 24000, 30001]
    g = torch.Generator().manual_seed(5)
    wavs = [0.1 * torch.randn(L, generator=g) for L in lens]
    x = torch.zeros(len(lens), max(lens))
    for i, w in enumerate(wavs):
        x[i, : w.numel()] = w
    with torch.no_grad():
        com, des = pair(x.cuda(), lengths=lens)
        for i, w in enumerate(wavs):
            c1, d1 = pair(w[None].cuda())
            torch.testing.assert_close(com[i], c1[0], rtol=2e-4, atol=2e-4)
            torch.testing.assert_close(des[i], d1[0], rtol=2e-4, atol=2e-4)
