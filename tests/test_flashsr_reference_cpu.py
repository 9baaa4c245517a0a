"""CPU: the PyTorch reference graph of the declared FlashSR table runs at the toy config, the UNet block list is
consistent (skip stack empties), the parameter table covers what the graph touches, filters are sane."""
import numpy as np
import torch


def test_reference_graph_toy(pack):
    from egregora_amd import flashsr_arch as A
    from oracle import flashsr_torch as R
    cfg = A.tiny_config()
    P = A.init_params(cfg, 0)
    g = torch.Generator().manual_seed(1)
    x = 0.3 * torch.randn(2, cfg.chunk, generator=g)
    h, w = cfg.lat_hw
    st = {}
    with torch.no_grad():
        y = R.flashsr_forward(x, torch.randn(2, cfg.z_ch, h, w, generator=g), P, cfg, A.unet_blocks(cfg),
                              torch.from_numpy(A.mel_filterbank(cfg)), torch.from_numpy(A.kaiser_sinc_filter(cfg.aa_taps)), st)
    assert y.shape == (2, cfg.chunk) and torch.isfinite(y).all()
    assert st["mel"].shape == (2, 1, cfg.n_frames, cfg.n_mels) and st["z_cond"].shape == (2, cfg.z_ch, h, w)
    assert float(y.abs().max()) <= 1.0


def test_full_table_shapes(pack):
    from egregora_amd import flashsr_arch as A
    cfg = A.FlashSRConfig()
    assert cfg.chunk == 245760 and cfg.chunk // cfg.hop == cfg.n_frames and cfg.lat_hw == (64, 32)
    assert int(np.prod(cfg.voc_rates)) == cfg.hop
    blocks = A.unet_blocks(cfg)
    depth = 0
    for name, cin, cout, attn in blocks:
        part, _, kind = name.split(".")
        if part == "in":
            depth += 1
        if part == "out" and kind == "block":
            depth -= 1
    assert depth == 0
    fb = A.mel_filterbank(cfg)
    assert fb.shape == (256, 1025) and (fb >= 0).all() and (fb.sum(1) > 0).all()
    f = A.kaiser_sinc_filter(12)
    assert abs(float(f.sum()) - 1) < 1e-6 and np.allclose(f, f[::-1])
    a, s = A.cosine_alpha_sigma(cfg, cfg.t_steps - 1)
    assert abs(a * a + s * s - 1) < 1e-12 and a < 0.01
    for r in cfg.voc_rates:
        assert (A.up_kernel(r) - r) % 2 == 0


def test_config_round_trips_through_parameter_shapes(pack):
    from egregora_amd import flashsr_arch as A
    cfg = A.tiny_config()
    assert A.config_from_params(A.init_params(cfg, 0), cfg) == cfg
