"""HuggingFace ``Wav2Vec2Model`` as an independent proxy for the XLS-R front-end (test infrastructure; build container only:
``transformers`` is not needed on the GPU box).  fairseq -- where the reference's front-end arithmetic lives (sslassist.py:25, 48) --
is absent from the reference tree and the image, so this proxy does NOT pin parity with the reference; it replaces "HIP vs the
builder's own restatement" by "HIP vs an independently written implementation of the same published architecture".
``hf_model(cfg, p)`` builds the HF module for an ``oracle.xlsr_ref.XlsrConfig`` and loads fairseq-named parameters ``p`` into it."""


def hf_model(cfg, p):
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    hc = Wav2Vec2Config(hidden_size=cfg.dim, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                        intermediate_size=cfg.ffn, feat_extract_norm="layer", conv_bias=True,
                        do_stable_layer_norm=True, num_conv_pos_embeddings=cfg.pos_k,
                        num_conv_pos_embedding_groups=cfg.pos_groups, hidden_dropout=0.0,
                        attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0,
                        layerdrop=0.0, apply_spec_augment=False)
    m = Wav2Vec2Model(hc).eval()
    sd = m.state_dict()
    ren = {}
    for i in range(7):
        a, b = "feature_extractor.conv_layers.%d" % i, "feature_extractor.conv_layers.%d" % i
        ren[a + ".conv.weight"] = b + ".0.weight"; ren[a + ".conv.bias"] = b + ".0.bias"
        ren[a + ".layer_norm.weight"] = b + ".2.1.weight"; ren[a + ".layer_norm.bias"] = b + ".2.1.bias"
    ren["feature_projection.layer_norm.weight"] = "layer_norm.weight"
    ren["feature_projection.layer_norm.bias"] = "layer_norm.bias"
    ren["feature_projection.projection.weight"] = "post_extract_proj.weight"
    ren["feature_projection.projection.bias"] = "post_extract_proj.bias"
    ren["encoder.pos_conv_embed.conv.bias"] = "encoder.pos_conv.0.bias"
    ren["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = "encoder.pos_conv.0.weight_g"
    ren["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = "encoder.pos_conv.0.weight_v"
    ren["encoder.layer_norm.weight"] = "encoder.layer_norm.weight"
    ren["encoder.layer_norm.bias"] = "encoder.layer_norm.bias"
    for i in range(cfg.layers):
        a, b = "encoder.layers.%d" % i, "encoder.layers.%d" % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            for w in ("weight", "bias"):
                ren["%s.attention.%s.%s" % (a, n, w)] = "%s.self_attn.%s.%s" % (b, n, w)
        for w in ("weight", "bias"):
            ren["%s.layer_norm.%s" % (a, w)] = "%s.self_attn_layer_norm.%s" % (b, w)
            ren["%s.feed_forward.intermediate_dense.%s" % (a, w)] = "%s.fc1.%s" % (b, w)
            ren["%s.feed_forward.output_dense.%s" % (a, w)] = "%s.fc2.%s" % (b, w)
            ren["%s.final_layer_norm.%s" % (a, w)] = "%s.final_layer_norm.%s" % (b, w)
    new = {}
    for k, v in sd.items():
        if k in ren:
            assert tuple(v.shape) == tuple(p[ren[k]].shape), (k, v.shape, p[ren[k]].shape)
            new[k] = p[ren[k]].clone()
        else:
            assert k == "masked_spec_embed", k
            new[k] = v
    assert len(set(ren.values())) == len(p), (len(ren), len(p))
    m.load_state_dict(new, strict=True)
    return m
