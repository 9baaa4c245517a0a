"""SURVEY.md section 8 row a6 on the CPU: discovery of the three upstream checkpoint files, torch.load(weights_only=True), the
data-driven key map (flashsr_keymap.json), the layer table derived from the checkpoint's own shapes, and loud failure.

The upstream files do not exist in the build image, so the test writes a synthetic checkpoint set in the UPSTREAM spelling the
shipped map covers (LDM AutoencoderKL / openaimodel UNet / BigVGAN generator names, weight-normalised vocoder convolutions,
wrapper dicts, buffers to ignore) and takes it through discovery -> read -> map -> derived config -> parameter table.  What this
pins is the loader's mechanics and the map's self-consistency; that upstream really uses these names is UNPINNED (flashsr_weights.py)."""
import pytest
import torch


@pytest.fixture(scope="module")
def tiny(pack):
    from egregora_amd import flashsr_arch as A
    cfg = A.tiny_config()
    return cfg, A.init_params(cfg, 3)


def write_upstream_set(d, sds):
    d.mkdir(parents=True, exist_ok=True)
    torch.save({"state_dict": sds["vae.pth"]}, d / "vae.pth")                       # wrapper dicts as trainers write them
    torch.save(sds["student_ldm.pth"], d / "student_ldm.pth")
    torch.save({"generator": sds["sr_vocoder.pth"]}, d / "sr_vocoder.pth")


def test_round_trip_through_discovery_map_and_derived_config(pack, tiny, tmp_path, monkeypatch):
    from egregora_amd import flashsr_weights as W
    cfg, P = tiny
    sds = W.export_upstream_named(P)
    assert any(k.endswith("weight_g") for k in sds["sr_vocoder.pth"]) and "model.diffusion_model.time_embed.0.weight" in sds["student_ldm.pth"]
    assert "encoder.conv_in.weight" in sds["vae.pth"] and "resblocks.1.activations.3.act.alpha" in sds["sr_vocoder.pth"]
    sds["sr_vocoder.pth"]["resblocks.0.activations.0.upsample.filter"] = torch.zeros(1, 1, 12)      # buffers the map ignores
    sds["student_ldm.pth"]["model.diffusion_model.input_blocks.1.0.in_layers.0.num_batches_tracked"] = torch.zeros(())
    ck = tmp_path / "ComfyUI" / "models" / "audio" / "flashsr"
    write_upstream_set(ck, sds)
    monkeypatch.setenv("EGREGORA_FLASHSR_CKPT_DIR", str(ck))
    assert W.discover() == ck
    import dataclasses
    from egregora_amd import flashsr_arch as A
    # base: the FULL-SIZE table's widths (all wrong for this checkpoint) + the few fields shapes cannot reveal
    base = dataclasses.replace(A.FlashSRConfig(), chunk=cfg.chunk, n_fft=cfg.n_fft, n_frames=cfg.n_frames, gn_groups=cfg.gn_groups,
                               head_dim=cfg.head_dim, hop=1)
    params, got_cfg, where = W.load(base=base)
    assert where == ck and got_cfg == cfg                       # every width / depth / rate re-derived from the shapes alone
    assert set(params) == set(P)
    for k in P:
        tol = 0 if not (k.startswith("voc.") and k.endswith(".weight")) else 1e-6      # weight-norm fold: g * v / ||v||
        assert (params[k] - P[k]).abs().max() <= tol * P[k].abs().max(), k


def test_both_reference_locations_are_searched(pack, monkeypatch, tmp_path):
    from egregora_amd import flashsr_weights as W
    monkeypatch.delenv("EGREGORA_FLASHSR_CKPT_DIR", raising=False)
    root = tmp_path / "top" / "ComfyUI" / "custom_nodes" / "pack"
    monkeypatch.setattr(W, "pack_root", lambda: root)
    cands = W.candidate_dirs()
    assert cands == [tmp_path / "top" / "ComfyUI" / "models" / "audio" / "flashsr",        # README.md:160 / install.py:10-13
                     tmp_path / "top" / "models" / "audio" / "flashsr"]                     # reference :28-30 as written (Q2)
    with pytest.raises(RuntimeError, match=r"FlashSR weights missing\. Place these in models/audio/flashsr: student_ldm\.pth, sr_vocoder\.pth, vae\.pth"):
        W.discover()
    (cands[1]).mkdir(parents=True)
    for f in W.FILES:
        torch.save({}, cands[1] / f)
    assert W.discover() == cands[1]
    (cands[0]).mkdir(parents=True)
    for f in W.FILES[:2]:
        torch.save({}, cands[0] / f)
    assert W.discover() == cands[1]                              # an incomplete directory does not shadow a complete one
    with pytest.raises(RuntimeError, match="vae.pth"):
        W.load(cands[0])


def test_unmapped_missing_and_mismatched_tensors_fail_loudly(pack, tiny):
    from egregora_amd import flashsr_weights as W
    cfg, P = tiny
    sds = W.export_upstream_named(P)
    sds["vae.pth"]["encoder.some_new_layer.weight"] = torch.zeros(3)
    sds["student_ldm.pth"]["model.diffusion_model.label_emb.0.weight"] = torch.zeros(4, 4)
    with pytest.raises(RuntimeError) as e:
        W.map_checkpoints(sds, base=cfg)
    msg = str(e.value)
    assert "label_emb.0.weight" in msg and "not in the layer table" in msg or "unmapped upstream tensor" in msg
    sds = W.export_upstream_named(P)
    del sds["vae.pth"]["decoder.conv_out.bias"]
    w = sds["student_ldm.pth"]["model.diffusion_model.out.2.weight"]
    sds["student_ldm.pth"]["model.diffusion_model.out.2.weight"] = torch.zeros(w.shape[0], w.shape[1], 5, 5)
    with pytest.raises(RuntimeError) as e:
        W.map_checkpoints(sds, base=cfg)
    msg = str(e.value)
    assert "vae.decoder.conv_out.bias" in msg and "has no upstream tensor" in msg
    assert "shape mismatch at unet.out_conv.weight" in msg and "flashsr_keymap.json" in msg


def test_linear_transformer_projections_are_accepted(pack, tiny):
    """`use_linear_in_transformer` checkpoints store proj_in / proj_out as [C, C]; the table's 1x1 convolutions take them."""
    from egregora_amd import flashsr_weights as W
    cfg, P = tiny
    sds = W.export_upstream_named(P)
    u = sds["student_ldm.pth"]
    for k in [k for k in u if k.endswith("proj_in.weight") or (k.endswith("proj_out.weight") and ".1." in k)]:
        u[k] = u[k][:, :, 0, 0].clone()
    params, got_cfg = W.map_checkpoints(sds, base=cfg)
    assert got_cfg == cfg and all(torch.equal(params[k], P[k]) for k in P if ".st.proj_" in k)


def test_full_size_table_derives_from_shapes(pack):
    """config_from_params inverts the declared full-size table (names + shapes only, no storage)."""
    from egregora_amd import flashsr_arch as A
    cfg = A.FlashSRConfig()
    assert A.config_from_params(A.init_params(cfg, 0, shapes_only=True)) == cfg


def test_node_refuses_to_run_without_weights(pack, monkeypatch, tmp_path):
    """Without checkpoints the engine raises the reference's sentence -- synthetic weights are never picked up by the node path."""
    from egregora_amd import flashsr_engine as E, flashsr_weights as W, native
    monkeypatch.setenv("EGREGORA_FLASHSR_CKPT_DIR", str(tmp_path / "nothing"))
    monkeypatch.setenv("EGREGORA_FLASHSR_SYNTHETIC", "1")
    monkeypatch.delenv("EGREGORA_FLASHSR_WEIGHTS", raising=False)
    monkeypatch.setattr(W, "pack_root", lambda: tmp_path / "a" / "b" / "c" / "d")
    monkeypatch.setattr(native, "require_device", lambda: "gfx950")
    E.set_engine(None)
    with pytest.raises(RuntimeError, match="FlashSR weights missing"):
        E.ensure_ready()
