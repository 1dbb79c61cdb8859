"""CLIP `encode_text` (third-party, absent from the reference tree, parity unpinned against openai/CLIP itself)
cross-checked against an INDEPENDENT public implementation of the same architecture: Hugging Face `transformers`
`CLIPTextModelWithProjection` (v5.15.0 in this image) with the SAME random weights mapped from the OpenAI parameter names the
reference's checkpoint surgery sees (`net.clip.transformer.resblocks.N.attn.in_proj_weight`, ...).  It pins the oracle's
restatement (`oracle/restate.py:clip_encode_text`, the checker of the HIP text tower in `tests/test_gpu_policy_parity.py`) on CPU:
token + positional embedding, 12 pre-norm residual blocks with causal attention and QuickGELU, final LayerNorm, pooling at
the EOT token (argmax id), projection."""
import pytest
import torch

from oracle import restate as R

transformers = pytest.importorskip("transformers")


def _random_clip(layers, width=512, heads=8, ctx=77, vocab=49408, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    p = "net.clip"
    sd = {p + ".token_embedding.weight": rn(vocab, width), p + ".positional_embedding": rn(ctx, width, std=0.01),
          p + ".ln_final.weight": 1 + rn(width, std=0.1), p + ".ln_final.bias": rn(width, std=0.05),
          p + ".text_projection": rn(width, width, std=width ** -0.5)}
    for i in range(layers):
        b = f"{p}.transformer.resblocks.{i}"
        sd.update({b + ".ln_1.weight": 1 + rn(width, std=0.1), b + ".ln_1.bias": rn(width, std=0.05),
                   b + ".ln_2.weight": 1 + rn(width, std=0.1), b + ".ln_2.bias": rn(width, std=0.05),
                   b + ".attn.in_proj_weight": rn(3 * width, width, std=width ** -0.5), b + ".attn.in_proj_bias": rn(3 * width),
                   b + ".attn.out_proj.weight": rn(width, width, std=width ** -0.5), b + ".attn.out_proj.bias": rn(width),
                   b + ".mlp.c_fc.weight": rn(4 * width, width, std=width ** -0.5), b + ".mlp.c_fc.bias": rn(4 * width),
                   b + ".mlp.c_proj.weight": rn(width, 4 * width, std=(4 * width) ** -0.5), b + ".mlp.c_proj.bias": rn(width)})
    return sd


def _to_hf(sd, layers, width=512):
    p, out = "net.clip", {}
    out["text_model.embeddings.token_embedding.weight"] = sd[p + ".token_embedding.weight"]
    out["text_model.embeddings.position_embedding.weight"] = sd[p + ".positional_embedding"]
    out["text_model.final_layer_norm.weight"] = sd[p + ".ln_final.weight"]
    out["text_model.final_layer_norm.bias"] = sd[p + ".ln_final.bias"]
    out["text_projection.weight"] = sd[p + ".text_projection"].t().contiguous()          # x @ P  ==  Linear(weight = P^T)
    for i in range(layers):
        b, h = f"{p}.transformer.resblocks.{i}", f"text_model.encoder.layers.{i}"
        w, bias = sd[b + ".attn.in_proj_weight"], sd[b + ".attn.in_proj_bias"]
        for j, name in enumerate(("q_proj", "k_proj", "v_proj")):
            out[f"{h}.self_attn.{name}.weight"] = w[j * width:(j + 1) * width]
            out[f"{h}.self_attn.{name}.bias"] = bias[j * width:(j + 1) * width]
        out[f"{h}.self_attn.out_proj.weight"] = sd[b + ".attn.out_proj.weight"]
        out[f"{h}.self_attn.out_proj.bias"] = sd[b + ".attn.out_proj.bias"]
        for ours, theirs in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            out[f"{h}.{theirs}.weight"] = sd[f"{b}.{ours}.weight"]
            out[f"{h}.{theirs}.bias"] = sd[f"{b}.{ours}.bias"]
    return out


def _tokens(lengths, seed=3):
    g = torch.Generator().manual_seed(seed)
    toks = torch.zeros(len(lengths), 77, dtype=torch.long)
    for b, ln in enumerate(lengths):
        toks[b, 0] = 49406
        toks[b, 1:ln] = torch.randint(1, 49406, (ln - 1,), generator=g)
        toks[b, ln] = 49407                                  # EOT: the largest id -> argmax pooling (CLIP) == first-EOS pooling (HF)
    return toks


@pytest.mark.parametrize("layers", [2, 12])
def test_oracle_clip_text_equals_huggingface_clip_text(layers):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    sd = _random_clip(layers)
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=512, intermediate_size=2048, projection_dim=512, num_hidden_layers=layers,
                         num_attention_heads=8, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                         pad_token_id=0, bos_token_id=49406, eos_token_id=49407, attention_dropout=0.0)
    hf = CLIPTextModelWithProjection(cfg).eval()
    missing, unexpected = hf.load_state_dict(_to_hf(sd, layers), strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    toks = _tokens([2, 3, 15, 16, 17, 33, 48, 64, 76])
    with torch.no_grad():
        ours = R.clip_encode_text(sd, "net.clip", toks)
        theirs = hf(input_ids=toks, attention_mask=None).text_embeds
    scale = float(theirs.abs().max())
    assert float((ours - theirs).abs().max()) < 2e-5 * max(scale, 1.0), float((ours - theirs).abs().max())
    # tokens behind the EOT cannot matter (causal attention + EOT pooling): what the HIP path's ragged batch relies on
    junk = toks.clone()
    for b in range(junk.shape[0]):
        e = int(toks[b].argmax())
        junk[b, e + 1:] = torch.randint(1, 49406, (76 - e,))
    with torch.no_grad():
        again = R.clip_encode_text(sd, "net.clip", junk)
    assert float((again - ours).abs().max()) < 1e-6
