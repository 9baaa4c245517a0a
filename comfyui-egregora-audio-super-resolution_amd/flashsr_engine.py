"""FlashSR on the MI355X.  Stands in for the reference's `_FlashSRRunner` (egregora_audio_super_resolution.py:254-369): built
once per process (the reference rebuilds per call, :393), inference on [rows, 245760] where rows = chunks x channels ride the
batch dimension (:366-368).

The model is the C-ABI handle -- `egr_flashsr_create` / `egr_flashsr_infer` (include/egregora_amd.h, csrc/egr_flashsr.cpp): the
graph walk, weight repacking and scratch arena live in the library; this module only hands it the named weight tensors
(`FlashSREngine.handle`, `infer_rows`) and shards chunk blocks -- over the ranks of a torch.distributed process group when one
exists (shard.py: one process per GPU, one all-gather), or, inside ONE process (what a ComfyUI host is), over the devices listed
in EGREGORA_DEVICES: one handle and one host thread per device, contiguous chunk blocks balanced to within one chunk, the gather
as peer copies into device 0's prediction tensor, WOLA on device 0 (`infer_spans_devices`).  A pass of >= 12 rows is split by the library into up
to EGREGORA_FLASHSR_STREAMS (default 2) row groups that run as concurrent forwards on side streams the handle has verified to
sit on other hardware queues; the caller sees one stream.

(The operator-by-operator Python walk of the same graph that the tests use for per-stage taps and ablations is NOT part of the
product: tools/flashsr_pydriver.py, a subclass of this engine, held bit-for-bit equal to the handle by
tests/test_gpu_flashsr_capi.py.)

Activations are channels-last ([B][H][W][C] / [B][L][C]) float32.
PARITY UNPINNED vs upstream (see flashsr_arch.py); checked against oracle/flashsr_torch.py (same table, torch fp32 / fp64).
"""
import ctypes as C
import math
import os
import threading
from typing import Dict, List, Optional, Tuple

import torch

from . import device_ops, flashsr_arch as arch, native, shard

ACT_NONE, ACT_SILU, ACT_TANH, ACT_LEAKY, ACT_LOGCLAMP = 0, 1, 2, 3, 4
EW_ADD, EW_AXPBY, EW_SILU, EW_SCALE, EW_COPY, EW_ADD_SCALE = 0, 1, 2, 3, 4, 5


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class FlashSREngine:
    # Switches of the handle (frozen per engine at construction; egr_flashsr_create flags, include/egregora_amd.h):
    # dense contractions run on the bf16 matrix pipe with fp32-grade results ("bf16x3": exact three-way split of both
    # operands, six partial products accumulated in fp32, csrc/egr_nn_gemm_s3.hip) or on v_mfma_f32_32x32x2_f32 ("f32").
    MFMA_MODE = os.environ.get("EGREGORA_FLASHSR_MFMA", "bf16x3")
    # operand scheme of egr_flashsr_infer's split contractions: "f16x2" = two fp16 terms, one power-of-two scale per (tensor, batch
    # row) derived on the device from that row's own maximum (include/egregora_amd.h egr_flashsr_set_split), "bf16x3" = always
    # three bf16 terms.  egr_flashsr_forward / the operator API use the bf16 terms.
    SPLIT = os.environ.get("EGREGORA_FLASHSR_SPLIT", "f16x2")
    # F(2x2,3x3) paid from 256 channels (its transforms move 4x the tensor); F(4x4,3x3) moves 2.25x and pays from 128
    WINO_MIN_CH = int(os.environ.get("EGREGORA_FLASHSR_WINOGRAD_MIN_CH", "128"))
    WINO_F4 = os.environ.get("EGREGORA_FLASHSR_WINOGRAD_F4", "1") != "0"   # F(4x4,3x3) where H and W are multiples of 4
    GN_PARTIALS = os.environ.get("EGREGORA_FLASHSR_GN_PARTIALS", "1") != "0"
    THIN_ENDS = os.environ.get("EGREGORA_FLASHSR_THIN_ENDS", "1") != "0"     # dedicated paths for Cout*kh*kw <= 32 and Cin == 1 convs
    FUSE_GN = os.environ.get("EGREGORA_FLASHSR_FUSE_GN", "1")      # "0" off, "1" Winograd consumers only, "all"
    _KEEP_PARAMS = False        # the test driver (tools/flashsr_pydriver.py) keeps the state dict for its own lazy packs

    def __init__(self, cfg: arch.FlashSRConfig, params: Dict[str, torch.Tensor], device="cuda"):
        native.require_device()
        self.cfg = cfg
        self.dev = torch.device(device)
        if self.dev.index is None:               # pinned to the device that is current now: the handle's memory lives there
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.L = native.lib()
        # the class-level switches are frozen per engine at construction: the handle flags must see the values this engine was
        # built under, whatever the class holds later
        for name in dir(type(self)):
            if name.isupper() and not name.startswith("_"):
                setattr(self, name, getattr(type(self), name))
        self.mfma = self.MFMA_MODE
        self.thin = self.THIN_ENDS
        self._handle = None
        self._params = {k: v.detach().float().contiguous() for k, v in params.items()}     # torch layouts (host)
        self.window = torch.hann_window(cfg.n_fft, periodic=True, dtype=torch.float32).to(self.dev)
        self.filt = torch.from_numpy(arch.kaiser_sinc_filter(cfg.aa_taps)).to(self.dev)
        self.mel_fb = torch.from_numpy(arch.mel_filterbank(cfg)).contiguous().to(self.dev)       # [n_mels][nb]

    def _st(self):
        return native.stream_ptr()

    # ------------------------------------------------------------------ the C-ABI model handle (product path)
    def time_embedding_input(self) -> torch.Tensor:
        """Sinusoidal embedding of the fixed step t = T-1, [1, unet_ch] float32 (the input of unet.time_embed)."""
        half = self.cfg.unet_ch // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        args = float(self.cfg.t_steps - 1) * freqs
        return torch.cat([torch.cos(args), torch.sin(args)])[None].contiguous().to(self.dev)

    def named_tensors(self) -> Dict[str, torch.Tensor]:
        """What egr_flashsr_create takes: the layer table's parameters in torch layouts + the four derived constants."""
        if self._params is None:
            raise RuntimeError("the engine's state dict was released after the handle was built")
        t = {k: v.to(self.dev) for k, v in self._params.items()}
        t["const.window"] = self.window
        t["const.mel_fb"] = self.mel_fb
        t["const.aa_filter"] = self.filt
        t["const.time_emb"] = self.time_embedding_input()
        return t

    def handle_flags(self) -> int:
        f = native.FSR_F32_MFMA if self.mfma != "bf16x3" else 0
        f |= native.FSR_NO_WINOGRAD if self.WINO_MIN_CH >= (1 << 20) else 0
        f |= 0 if self.WINO_F4 else native.FSR_NO_WINO_F4
        f |= 0 if self.GN_PARTIALS else native.FSR_NO_GN_PARTIALS
        f |= 0 if self.thin else native.FSR_NO_THIN_ENDS
        f |= native.FSR_NO_FUSE_GN if self.FUSE_GN == "0" else 0
        f |= native.FSR_SPLIT_BF16X3 if self.SPLIT == "bf16x3" else 0
        return f

    @property
    def handle(self) -> int:
        """egr_flashsr handle built from this engine's parameters (once)."""
        if self._handle is None:
            with torch.cuda.device(self.dev):      # egr_flashsr_create allocates on the CURRENT device
                named = self.named_tensors()
                descs = (native.TensorDescC * len(named))()
                keep = []
                for d, (k, v) in zip(descs, named.items()):
                    v = v.contiguous()
                    keep.append(v)
                    d.name, d.data, d.ndim = k.encode(), v.data_ptr(), v.dim()
                    for i, n in enumerate(v.shape):
                        d.shape[i] = n
                cc = native.flashsr_config_c(self.cfg)
                out = C.c_void_p()
                native.check(self.L.egr_flashsr_create(C.byref(out), C.byref(cc), descs, len(named), self.handle_flags(), self._st()),
                             "egr_flashsr_create")
                self._handle = out.value
                if not self._KEEP_PARAMS:
                    self._params = None          # the library holds its own packed copy: drop the host-side state dict
                native.check(self.L.egr_flashsr_set_rows_per_pass(C.c_void_p(self._handle), ROWS_PER_PASS), "egr_flashsr_set_rows_per_pass")
        return self._handle

    def close(self):
        if self._handle is not None:
            self.L.egr_flashsr_destroy(C.c_void_p(self._handle))
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    def c_forward(self, x: torch.Tensor, noise: torch.Tensor, stages: Optional[dict] = None, lowpass: bool = False) -> torch.Tensor:
        """egr_flashsr_forward: one pass through the library's graph walk; same arguments and results as forward_rows."""
        x = x.contiguous()
        R = x.shape[0]
        cfg = self.cfg
        y = torch.empty((R, cfg.chunk), dtype=torch.float32, device=self.dev)
        ptrs = None
        if stages is not None:
            h, w = cfg.lat_hw
            bufs = dict(mel=torch.empty((R, cfg.n_frames, cfg.n_mels, 1), dtype=torch.float32, device=self.dev),
                        z_cond=torch.empty((R, h, w, cfg.z_ch), dtype=torch.float32, device=self.dev),
                        v=torch.empty((R, h, w, cfg.z_ch), dtype=torch.float32, device=self.dev),
                        z0=torch.empty((R, h, w, cfg.z_ch), dtype=torch.float32, device=self.dev),
                        mel_hat=torch.empty((R, cfg.n_frames, cfg.n_mels, 1), dtype=torch.float32, device=self.dev),
                        y=torch.empty((R, cfg.n_frames * cfg.hop), dtype=torch.float32, device=self.dev))
            ptrs = (C.c_void_p * 6)(*[bufs[k].data_ptr() for k in ("mel", "z_cond", "v", "z0", "mel_hat", "y")])
            stages.update(bufs)
        native.check(self.L.egr_flashsr_forward(C.c_void_p(self.handle), _p(x), _p(noise.contiguous()), R, 1 if lowpass else 0, _p(y), ptrs,
                                                self._st()), "egr_flashsr_forward")
        return y

    def c_infer(self, x: torch.Tensor, row_ids: Optional[torch.Tensor], seed: int, lowpass: bool = False) -> torch.Tensor:
        """egr_flashsr_infer: [R, chunk] -> [R, chunk], noise keyed by (seed, global row id), rows-per-pass handled in the library."""
        x = x.contiguous()
        y = torch.empty_like(x)
        native.check(self.L.egr_flashsr_set_rows_per_pass(C.c_void_p(self.handle), ROWS_PER_PASS), "egr_flashsr_set_rows_per_pass")
        native.check(self.L.egr_flashsr_infer(C.c_void_p(self.handle), _p(x), x.shape[0], 1 if lowpass else 0, int(seed) & (2 ** 64 - 1),
                                              _p(row_ids.contiguous()) if row_ids is not None else None, _p(y), self._st()),
                     "egr_flashsr_infer")
        return y

    def warmup(self, rows: int = 2):
        """egr_flashsr_warmup: one throw-away pass over `rows` rows of silence (default: one stereo chunk) -- the kernels' code objects
        are loaded and the scratch for that row count exists before the host's first real call.  EGREGORA_FLASHSR_WARMUP=0 skips it."""
        if os.environ.get("EGREGORA_FLASHSR_WARMUP", "1") == "0" or rows < 1:
            return
        with torch.cuda.device(self.dev):
            native.check(self.L.egr_flashsr_warmup(C.c_void_p(self.handle), int(rows), self._st()), "egr_flashsr_warmup")

    def set_split(self, scheme: str):
        """"bf16x3" or "f16x2" for the following c_infer calls (the latter needs an engine built with SPLIT = "f16x2")."""
        code = {"bf16x3": 0, "f16x2": 1, "f16x2+forward": 2}[scheme]     # the last: c_forward too runs the fp16 operand terms (tests)
        native.check(self.L.egr_flashsr_set_split(C.c_void_p(self.handle), code), "egr_flashsr_set_split")

    def set_arena_cap_gb(self, gb: float):
        """Scratch budget of the handle in GB (0: none; EGREGORA_FLASHSR_ARENA_GB sets it at creation): fewer rows per pass instead of
        an allocation failure next to the host's other models (include/egregora_amd.h egr_flashsr_set_arena_cap)."""
        native.check(self.L.egr_flashsr_set_arena_cap(C.c_void_p(self.handle), float(gb) * 1e9), "egr_flashsr_set_arena_cap")

    def split_info(self) -> dict:
        """enabled: c_infer runs two fp16 terms per operand with per-row scales derived on the device; weights: contraction weights that
        hold fp16 terms; calls: c_infer calls made on the scheme (include/egregora_amd.h egr_flashsr_set_split)."""
        en, nw, calls = C.c_int(), C.c_int(), C.c_int64()
        native.check(self.L.egr_flashsr_split_info(C.c_void_p(self.handle), C.byref(en), C.byref(nw), C.byref(calls)), "egr_flashsr_split_info")
        return dict(enabled=bool(en.value), weights=nw.value, calls=calls.value)

    def c_profile(self, fn) -> dict:
        """{kernel instantiation: (launches, flops, ms)} of the MFMA contraction launches the handle made while fn() ran."""
        h = C.c_void_p(self.handle)
        native.check(self.L.egr_flashsr_set_profiling(h, 1), "egr_flashsr_set_profiling")
        fn()
        n = C.c_int()
        native.check(self.L.egr_flashsr_profile(h, 0, None, 0, None, None, None, C.byref(n)), "egr_flashsr_profile")
        out = {}
        for i in range(n.value):
            buf, la, fl, ms = C.create_string_buffer(128), C.c_int64(), C.c_double(), C.c_double()
            native.check(self.L.egr_flashsr_profile(h, i, buf, 128, C.byref(la), C.byref(fl), C.byref(ms), None), "egr_flashsr_profile")
            out[buf.value.decode()] = (la.value, fl.value, ms.value)
        native.check(self.L.egr_flashsr_set_profiling(h, 0), "egr_flashsr_set_profiling")
        return out

    def noise(self, rows: int, row_ids: Optional[torch.Tensor], seed: int) -> torch.Tensor:
        """The library's Philox noise [rows, h, w, z] keyed by (seed, global row id): what egr_flashsr_infer draws internally."""
        h, w = self.cfg.lat_hw
        out = torch.empty((rows, h, w, self.cfg.z_ch), dtype=torch.float32, device=self.dev)
        native.check(self.L.egr_randn(_p(out), h * w * self.cfg.z_ch, rows, int(seed) & (2 ** 64 - 1), _p(row_ids),
                                      self._st()), "egr_randn")
        return out

    def flop_count(self, rows: int = 1) -> float:
        """Dense-contraction flops of one forward over `rows` rows, counted by the library's graph walk."""
        fl = C.c_double()
        native.check(self.L.egr_flashsr_flop_count(C.c_void_p(self.handle), rows, C.byref(fl), None), "egr_flashsr_flop_count")
        return fl.value


# ---------------------------------------------------------------------------------------------------- module state
# _ENGINES: device index -> engine (one C handle per device), plus ("slot", i) -> engine for position i of EGREGORA_DEVICES when a
# device index repeats there ("0,0": two handles on one GPU -- a handle serves one host thread at a time).  Every read-modify-write
# of the two globals happens under _LOCK; worker threads never build engines (infer_spans_devices resolves them on the caller's
# thread before it starts the workers), so the checkpoints are read ONCE per process whatever the device count.
_ENGINES: Dict[object, FlashSREngine] = {}
_SOURCE: Optional[Tuple[arch.FlashSRConfig, Dict[str, torch.Tensor]]] = None    # host state dict, kept only while more devices may need a copy
_LOCK = threading.RLock()
ROWS_PER_PASS = int(os.environ.get("EGREGORA_FLASHSR_ROWS", "32"))
SEED = int(os.environ.get("EGREGORA_FLASHSR_SEED", "0"))
# rows of the throw-away pass an engine runs when it is built (FlashSREngine.warmup): one stereo chunk loads every kernel the graph
# launches at small row counts; EGREGORA_FLASHSR_WARMUP_ROWS=26 also sizes the arenas of a 60 s stereo file, 0 skips the pass
WARMUP_ROWS = int(os.environ.get("EGREGORA_FLASHSR_WARMUP_ROWS", "2"))


def devices() -> List[int]:
    """Devices the node shards its chunks over inside THIS process: EGREGORA_DEVICES = "0,1,2,..." or "all" (default: the current
    device only).  The first entry stitches (WOLA).  An index may repeat ("0,0": two handles and two host threads on one GPU --
    the single-GPU test of the multi-device path)."""
    spec = os.environ.get("EGREGORA_DEVICES", "").strip()
    if not spec:
        return [torch.cuda.current_device()]
    if spec.lower() == "all":
        return list(range(torch.cuda.device_count()))
    try:
        devs = [int(t) for t in spec.split(",") if t.strip() != ""]
    except ValueError:
        raise RuntimeError(f"EGREGORA_DEVICES={spec!r}: expected a comma-separated list of device indices or 'all'")
    n = torch.cuda.device_count()
    bad = [d for d in devs if not 0 <= d < n]
    if bad or not devs:
        raise RuntimeError(f"EGREGORA_DEVICES={spec!r}: device(s) {bad} not among the {n} visible")
    return devs


def _load_source():
    from . import flashsr_weights
    path = os.environ.get("EGREGORA_FLASHSR_WEIGHTS", "")
    if path:
        if not os.path.exists(path):
            raise RuntimeError(f"EGREGORA_FLASHSR_WEIGHTS={path} (missing).")
        params = torch.load(path, map_location="cpu", weights_only=True)
        return arch.config_from_params(params), params
    params, cfg, where = flashsr_weights.load()
    print(f"[FlashSR] checkpoints from {where}: {len(params)} tensors, layer table {cfg}")
    return cfg, params


def _build_engine(cfg, params, dev: int) -> "FlashSREngine":
    """One engine on `dev` from the host state dict (a separate function so that the tests can count / slow down builds)."""
    eng = FlashSREngine(cfg, params, device=f"cuda:{dev}")
    eng.handle                                               # egr_flashsr_create now, on this thread: nothing is left for a worker to build
    eng.warmup(WARMUP_ROWS)
    return eng


def _slot_keys(devs: List[int]) -> List[object]:
    """Registry key of every position of a device list: the device index the first time it appears, ("slot", i) for a repeat --
    "0,0" is two handles (a handle serves one host thread at a time), "0,1,2" is one per device."""
    seen, keys = set(), []
    for i, d in enumerate(devs):
        keys.append(d if d not in seen else ("slot", i))
        seen.add(d)
    return keys


def _ensure(keys_devs: List[Tuple[object, int]]) -> List["FlashSREngine"]:
    """Engines for (registry key, device) pairs, built serially under the module lock from ONE load of the checkpoints."""
    global _SOURCE
    with _LOCK:
        missing = [(k, d) for k, d in keys_devs if k not in _ENGINES]
        if missing:
            native.require_device()
            if _SOURCE is None:
                _SOURCE = _load_source()
            cfg, params = _SOURCE
            for k, d in missing:
                if k not in _ENGINES:                        # a key may repeat in the request
                    _ENGINES[k] = _build_engine(cfg, params, d)
            if all(k in _ENGINES for k in _slot_keys(devices())):
                _SOURCE = None                               # every handle that will be used holds its packed copy: drop the host state dict
        return [_ENGINES[k] for k, _ in keys_devs]


def ensure_ready(device: Optional[int] = None) -> FlashSREngine:
    """Build the engine of `device` (default: the current one) once per process (the reference rebuilds the model on every run(),
    :393).  Weight sources, in order:
      1. EGREGORA_FLASHSR_WEIGHTS -- a torch state dict already in this pack's layer-table names (flashsr_arch.py);
      2. the three upstream checkpoints student_ldm.pth / sr_vocoder.pth / vae.pth discovered in models/audio/flashsr
         (flashsr_weights.discover: both places the reference can mean, or EGREGORA_FLASHSR_CKPT_DIR), mapped through
         flashsr_keymap.json and validated tensor by tensor;
      3. otherwise the reference's "weights missing" error (:314-317).
    The layer table always comes from the checkpoint's own tensor shapes (flashsr_arch.config_from_params).  Seeded synthetic
    weights are for benches and tests only and are never picked up here: those callers build a FlashSREngine themselves and
    install it with set_engine().  Thread-safe: concurrent callers serialise on the module lock and the checkpoints are read once."""
    dev = torch.cuda.current_device() if device is None and torch.cuda.is_available() else (device or 0)
    eng = _ENGINES.get(dev)                                  # (a plain dict read: safe next to a locked writer under the GIL)
    return eng if eng is not None else _ensure([(dev, dev)])[0]


def engines_for(devs: List[int]) -> List[FlashSREngine]:
    """One engine per POSITION of `devs` (a repeated device index gets its own handle), resolved or built on the calling thread."""
    return _ensure(list(zip(_slot_keys(devs), devs)))


def set_engine(engine: Optional[FlashSREngine]):
    """Install a prebuilt engine for its device (benches, tests); None forgets every engine."""
    global _SOURCE
    with _LOCK:
        if engine is None:
            _ENGINES.clear()
            _SOURCE = None
        else:
            _ENGINES[engine.dev.index] = engine


def set_engines(engines: List[FlashSREngine], devs: Optional[List[int]] = None):
    """Install one prebuilt engine per entry of `devs` (default: each engine's own device), keyed as engines_for() looks them up."""
    devs = [e.dev.index for e in engines] if devs is None else list(devs)
    with _LOCK:
        _ENGINES.clear()
        for k, e in zip(_slot_keys(devs), engines):
            _ENGINES[k] = e


def infer_rows(eng: FlashSREngine, rows_x: torch.Tensor, row_ids: torch.Tensor, seed: int,
               lowpass: bool = False) -> torch.Tensor:
    """rows_x [R, chunk] -> [R, chunk] through the C-ABI handle (egr_flashsr_infer: ROWS_PER_PASS rows at a time on the caller's
    stream, noise keyed by global row id; concurrent row groups inside the call, see the module docstring)."""
    return eng.c_infer(rows_x, row_ids, seed, lowpass)


def balanced_bounds(n: int, k: int) -> List[Tuple[int, int]]:
    """n chunks over k workers as contiguous blocks whose sizes differ by at most one (130 over 8: 17, 17, 16, 16, 16, 16, 16, 16)."""
    base, extra = divmod(n, k) if k > 0 else (0, 0)
    out, lo = [], 0
    for i in range(k):
        hi = lo + base + (1 if i < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def infer_spans(x_ct: torch.Tensor, n_chunks: int, win: int, hop: int, lowpass: bool) -> torch.Tensor:
    """[C,T] @48 kHz on the GPU -> predictions [n_chunks, C, win].  Chunks are sharded over the ranks of the default process
    group when one exists (contiguous blocks, one all-gather; shard.py); otherwise over the devices of EGREGORA_DEVICES inside
    this process (infer_spans_devices); otherwise they all run here."""
    Cn = x_ct.shape[0]
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return shard.sharded_chunks(lambda lo, hi: infer_block(x_ct, lo, hi, win, hop, lowpass), n_chunks, (Cn, win), x_ct.device)
    devs = devices() if x_ct.is_cuda else []
    if len(devs) > 1 and n_chunks > 1:
        return infer_spans_devices(x_ct, n_chunks, win, hop, lowpass, devs)
    return infer_block(x_ct, 0, n_chunks, win, hop, lowpass)


def infer_spans_devices(x_ct: torch.Tensor, n_chunks: int, win: int, hop: int, lowpass: bool, devs: List[int]) -> torch.Tensor:
    """The reference's chunk loop (egregora_audio_super_resolution.py:407-420) spread over several GPUs of ONE process: worker i
    (a host thread bound to devs[i], with that device's own handle) takes the contiguous block balanced_bounds(n)[i], reads the
    input through a peer copy, runs its rows batched, and writes its predictions straight into its slice of the [n, C, win]
    tensor on x_ct's device (peer copy = the gather); the caller stitches there.  Noise is keyed by the GLOBAL row id and every
    row's result depends on that row alone, so the outcome does not depend on the device count beyond fp32 round-off from tile
    choices (they follow the row count of a forward).  UNMEASURED on more than one physical GPU in this repository's test
    pool: the one-GPU box exercises it as EGREGORA_DEVICES=0,0."""
    Cn = x_ct.shape[0]
    home = x_ct.device
    out = torch.empty((n_chunks, Cn, win), dtype=torch.float32, device=home)
    torch.cuda.current_stream(home).synchronize()            # x_ct is complete before another device's stream reads it
    engs = engines_for(devs)                                 # built here, serially, from one load of the checkpoints: the workers only run
    jobs = [(engs[i], d, lo, hi) for i, (d, (lo, hi)) in enumerate(zip(devs, balanced_bounds(n_chunks, len(devs)))) if hi > lo]
    errors: List[BaseException] = []

    def work(eng: FlashSREngine, d: int, lo: int, hi: int):
        try:
            with torch.cuda.device(d):
                xd = x_ct if torch.device("cuda", d) == home else x_ct.to(torch.device("cuda", d), non_blocking=True)
                preds = infer_block(xd, lo, hi, win, hop, lowpass, eng)
                out[lo:hi].copy_(preds, non_blocking=True)   # the gather: a peer copy into the stitching device's tensor
                torch.cuda.current_stream().synchronize()
        except BaseException as ex:      # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(ex)

    threads = [threading.Thread(target=work, args=j, name=f"flashsr-dev{j[1]}") for j in jobs[1:]]
    for t in threads:
        t.start()
    work(*jobs[0])                                           # the first block runs on the caller's thread
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def infer_block(x_ct: torch.Tensor, lo: int, hi: int, win: int, hop: int, lowpass: bool, eng: Optional[FlashSREngine] = None) -> torch.Tensor:
    """Predictions [hi - lo, C, win] of the chunks lo .. hi-1 of x_ct (one rank's or device's share; noise keyed by the GLOBAL row id)."""
    eng = eng or ensure_ready(x_ct.device.index)
    Cn = x_ct.shape[0]
    if win != eng.cfg.chunk:
        raise RuntimeError(f"chunk length {win} != model chunk {eng.cfg.chunk}")
    chunks = device_ops.chunk_gather(x_ct, win, hop, lo, hi - lo)                 # [n, C, win]
    ids = (torch.arange(lo, hi, device=x_ct.device, dtype=torch.int64)[:, None] * Cn +
           torch.arange(Cn, device=x_ct.device, dtype=torch.int64)[None, :]).reshape(-1)
    y = infer_rows(eng, chunks.view(-1, win), ids, SEED, lowpass=bool(lowpass))
    return y.view(hi - lo, Cn, win)
