"""The Fat-Llama nodes' three input modes on the device (reference egregora_fat_llama_gpu.py:40-80): AUDIO dict, audio_path,
audio_url -- WAV and FLAC containers -- and `target_format` "flac".  The path / url forms pass through `_to_cs` (peak > 1 rescale),
the dict form does not; a file holds PCM, so all three must give the same node output for the same samples."""
import functools
import http.server
import threading

import numpy as np
import pytest
import torch

from flac_encoder import encode
from oracle import fatllama as ofl

pytestmark = pytest.mark.gpu


def signal():
    rng = np.random.Generator(np.random.PCG64(21))
    t = np.arange(12000) / 48000.0
    x = np.stack([0.4 * np.sin(2 * np.pi * 330 * t), 0.3 * np.sin(2 * np.pi * 1200 * t + 1)]) + 0.01 * rng.standard_normal((2, 12000))
    q = np.clip(np.rint(x * 32767.0), -32768, 32767).astype(np.int64)
    return q, (q / 32768.0).astype(np.float32)                     # PCM_16 integers and what sf.read returns for them


@pytest.mark.parametrize("node_key,extra", [("EgregoraFatLlamaGPU", (True, True)), ("EgregoraFatLlamaCPU", ())])
def test_dict_path_and_url_inputs_agree(pack, tmp_path, node_key, extra):
    from egregora_amd import wavio
    q, xf = signal()
    (tmp_path / "in.flac").write_bytes(encode(q, 48000, stereo="mid_side", plan=lambda fi, c: dict(kind=("fixed", 2), porder=2)))
    wavio.write_wav_pcm16(str(tmp_path / "in.wav"), (q.T / 32767.0).astype(np.float32), 48000)      # rint(x * 32767) gives q back
    node = pack.NODE_CLASS_MAPPINGS[node_key]()
    args = ("flac", 12, 0.6, 1536) + extra
    (ref,) = node.run(*args, AUDIO={"waveform": torch.from_numpy(xf)[None], "sample_rate": 48000})
    want, sr = ofl.node_run(xf, 48000, 12, 0.6, 1536, True, True)
    assert ref["sample_rate"] == sr == 48000
    lsb = np.abs(ref["waveform"][0].numpy() - want) * 32768.0
    assert float(lsb.max()) <= 1.0 + 1e-6 and float(np.mean(lsb > 0.5)) <= 5e-2
    for name in ("in.flac", "in.wav"):
        (got,) = node.run(*args, audio_path=str(tmp_path / name))
        assert got["sample_rate"] == 48000 and torch.equal(got["waveform"], ref["waveform"]), name
    handler = functools.partial(http.server.SimpleHTTPRequestHandler, directory=str(tmp_path))
    srv = http.server.ThreadingHTTPServer(("127.0.0.1", 0), handler)
    th = threading.Thread(target=srv.serve_forever, daemon=True)
    th.start()
    try:
        (got,) = node.run(*args, audio_url=f"http://127.0.0.1:{srv.server_address[1]}/in.flac")
        assert torch.equal(got["waveform"], ref["waveform"])
    finally:
        srv.shutdown()
    with pytest.raises(RuntimeError, match="audio_path not found"):
        node.run(*args, audio_path=str(tmp_path / "missing.flac"))
    with pytest.raises(RuntimeError, match="No AUDIO provided"):
        node.run(*args)
