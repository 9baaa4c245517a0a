#!/usr/bin/env python3
"""Writes comfyui-egregora-audio-super-resolution_amd/flashsr_keymap.json: the data-driven map between upstream FlashSR checkpoint tensor
names and this pack's layer-table names (flashsr_arch.py), in both directions.

The upstream spellings are those of the published modules FlashSR is assembled from, as recalled (UPSTREAM-RECALL -- the upstream
repository is absent from the build image): LDM `AutoencoderKL`, LDM `openaimodel.UNetModel` + `SpatialTransformer`, BigVGAN's
generator with weight-normalised convolutions.  flashsr_weights.map_checkpoints refuses anything the map does not cover, listing
the tensors, so a wrong guess here is loud and is fixed by editing the JSON (or this generator) -- never the kernels."""
import json
from pathlib import Path

WB = r"(weight|bias)"
RES = [("in_layers.0", "res.in_norm"), ("in_layers.2", "res.in_conv"), ("emb_layers.1", "res.emb"), ("out_layers.0", "res.out_norm"),
       ("out_layers.3", "res.out_conv"), ("skip_connection", "res.skip")]
ST = [("norm", "st.norm"), ("proj_in", "st.proj_in"), ("proj_out", "st.proj_out"),
      ("transformer_blocks.0.norm1", "st.attn1_ln"), ("transformer_blocks.0.norm2", "st.attn2_ln"), ("transformer_blocks.0.norm3", "st.ff_ln"),
      ("transformer_blocks.0.attn1.to_out.0", "st.attn1.to_out"), ("transformer_blocks.0.attn2.to_out.0", "st.attn2.to_out"),
      ("transformer_blocks.0.ff.net.0.proj", "st.ff.geglu"), ("transformer_blocks.0.ff.net.2", "st.ff.out")]
QKV = [(f"transformer_blocks.0.attn{a}.to_{p}", f"st.attn{a}.to_{p}") for a in (1, 2) for p in "qkv"]
esc = lambda s: s.replace(".", r"\.")


def unet_rules():
    fwd, inv = [], []
    fwd.append([rf"^time_embed\.(0|2)\.{WB}$", "unet.time_embed.{g1}.{g2}"])
    inv.append([rf"^unet\.time_embed\.(0|2)\.{WB}$", "time_embed.{g1}.{g2}"])
    fwd.append([rf"^input_blocks\.0\.0\.{WB}$", "unet.in.0.conv_in.{g1}"])
    inv.append([rf"^unet\.in\.0\.conv_in\.{WB}$", "input_blocks.0.0.{g1}"])
    fwd.append([rf"^input_blocks\.(\d+)\.0\.op\.{WB}$", "unet.in.{g1}.down.conv.{g2}"])
    inv.append([rf"^unet\.in\.(\d+)\.down\.conv\.{WB}$", "input_blocks.{g1}.0.op.{g2}"])
    fwd.append([rf"^output_blocks\.(\d+)\.[12]\.conv\.{WB}$", "unet.out.{g1}.up.conv.{g2}"])
    # export: the upsampler follows the transformer when the block has one; the exporter cannot know, so it writes slot 2 and the
    # forward rule accepts 1 or 2
    inv.append([rf"^unet\.out\.(\d+)\.up\.conv\.{WB}$", "output_blocks.{g1}.2.conv.{g2}"])
    places = [(r"input_blocks\.(\d+)", "unet.in.{g1}.block", r"unet\.in\.(\d+)\.block", "input_blocks.{g1}", 1),
              (r"middle_block", "unet.mid.0.block", r"unet\.mid\.0\.block", "middle_block", 0),
              (r"output_blocks\.(\d+)", "unet.out.{g1}.block", r"unet\.out\.(\d+)\.block", "output_blocks.{g1}", 1)]
    for up_rx, tb, tb_rx, up, ng in places:
        for a, b in RES:
            fwd.append([rf"^{up_rx}\.0\.{esc(a)}\.{WB}$", f"{tb}.{b}.{{g{ng + 1}}}"])
            inv.append([rf"^{tb_rx}\.{esc(b)}\.{WB}$", f"{up}.0.{a}.{{g{ng + 1}}}"])
        for a, b in ST:
            fwd.append([rf"^{up_rx}\.1\.{esc(a)}\.{WB}$", f"{tb}.{b}.{{g{ng + 1}}}"])
            inv.append([rf"^{tb_rx}\.{esc(b)}\.{WB}$", f"{up}.1.{a}.{{g{ng + 1}}}"])
        for a, b in QKV:
            fwd.append([rf"^{up_rx}\.1\.{esc(a)}\.weight$", f"{tb}.{b}.weight"])
            inv.append([rf"^{tb_rx}\.{esc(b)}\.weight$", f"{up}.1.{a}.weight"])
    # LDM's middle block is (res, attention, res); the table's mid.0 is res + attention, mid.1 the second res
    for a, b in RES:
        fwd.append([rf"^middle_block\.2\.{esc(a)}\.{WB}$", f"unet.mid.1.block.{b}.{{g1}}"])
        inv.insert(0, [rf"^unet\.mid\.1\.block\.{esc(b)}\.{WB}$", f"middle_block.2.{a}.{{g1}}"])
    fwd.append([rf"^out\.0\.{WB}$", "unet.out_norm.{g1}"])
    fwd.append([rf"^out\.2\.{WB}$", "unet.out_conv.{g1}"])
    inv.append([rf"^unet\.out_norm\.{WB}$", "out.0.{g1}"])
    inv.append([rf"^unet\.out_conv\.{WB}$", "out.2.{g1}"])
    return fwd, inv


def voc_rules():
    fwd = [[rf"^conv_pre\.{WB}$", "voc.conv_pre.{g1}"],
           [rf"^ups\.(\d+)\.0\.{WB}$", "voc.ups.{g1}.{g2}"],
           [rf"^resblocks\.(\d+)\.convs([12])\.(\d+)\.{WB}$", "voc.amp.{g1//nk}.{g1%nk}.{g3}.conv{g2}.{g4}"],
           [r"^resblocks\.(\d+)\.activations\.(\d+)\.act\.(alpha|beta)$", "voc.amp.{g1//nk}.{g1%nk}.{g2//2}.{g3}{g2%2+1}"],
           [r"^activation_post\.act\.(alpha|beta)$", "voc.post.{g1}"],
           [rf"^conv_post\.{WB}$", "voc.conv_post.{g1}"],
           # FlashSR's encoder of the low-resolution waveform (strided convolutions added U-Net style): spelling unverified
           [rf"^(?:wave_enc|audio_encoder|lr_encoder|cond_convs|downs)\.(\d+)(?:\.0)?\.{WB}$", "voc.wave_enc.{g1}.{g2}"]]
    inv = [[rf"^voc\.conv_pre\.{WB}$", "conv_pre.{g1}"],
           [rf"^voc\.ups\.(\d+)\.{WB}$", "ups.{g1}.0.{g2}"],
           [rf"^voc\.amp\.(\d+)\.(\d+)\.(\d+)\.conv([12])\.{WB}$", "resblocks.{g1*nk+g2}.convs{g4}.{g3}.{g5}"],
           [r"^voc\.amp\.(\d+)\.(\d+)\.(\d+)\.(alpha|beta)([12])$", "resblocks.{g1*nk+g2}.activations.{2*g3+g5-1}.act.{g4}"],
           [r"^voc\.post\.(alpha|beta)$", "activation_post.act.{g1}"],
           [rf"^voc\.conv_post\.{WB}$", "conv_post.{g1}"],
           [rf"^voc\.wave_enc\.(\d+)\.{WB}$", "wave_enc.{g1}.{g2}"]]
    return fwd, inv


def main():
    uf, ui = unet_rules()
    vf, vi = voc_rules()
    spec = {
        "_comment": "generated by tools/make_flashsr_keymap.py; edit there (or here) when upstream spells a tensor differently",
        "variables": {"int": "int"},
        "files": {
            "vae.pth": {"strip": ["first_stage_model.", "autoencoder.", "vae.", "module."],
                        "ignore": [r"^loss\.", r"^discriminator\.", r"num_batches_tracked$", r"^logvar$"],
                        "rules": [[r"^(encoder|decoder|quant_conv|post_quant_conv)\.(.+)$", "vae.{g1}.{g2}"]]},
            "student_ldm.pth": {"strip": ["model.diffusion_model.", "diffusion_model.", "student.", "unet.", "module."],
                                "ignore": [r"num_batches_tracked$", r"^(betas|alphas_cumprod|sqrt_|log_one_minus|posterior_|logvar)",
                                           r"^first_stage_model\.", r"^cond_stage_model\."],
                                "rules": uf,
                                "reshape": {r"^unet\..*\.st\.proj_(in|out)\.weight$": "unsqueeze_hw"}},
            "sr_vocoder.pth": {"strip": ["generator.", "vocoder.", "module."],
                               "ignore": [r"\.filter$", r"num_batches_tracked$"],
                               "rules": vf},
        },
        "export": {
            "vae.pth": {"prefix": "", "rules": [[r"^vae\.(.+)$", "{g1}"]]},
            "student_ldm.pth": {"prefix": "model.diffusion_model.", "rules": ui},
            "sr_vocoder.pth": {"prefix": "", "weight_norm": True, "rules": vi},
        },
    }
    spec["variables"] = {}
    out = Path(__file__).resolve().parent.parent / "comfyui-egregora-audio-super-resolution_amd" / "flashsr_keymap.json"
    out.write_text(json.dumps(spec, indent=1) + "\n", encoding="utf-8")
    print(out, sum(len(f["rules"]) for f in spec["files"].values()), "forward rules")


if __name__ == "__main__":
    main()
