"""BASELINE.json configs[3] and configs[4] at their stated sizes (one GPU's share), through size-independent properties:
generated payload -> channel -> receive round trip, determinism (a checksum of all outputs), iteration-count
invariants. Inputs are born in HBM (288 GB per MI355X holds them comfortably)."""
import numpy as np
import pytest

from conftest import SEED
from oraclelib import noise_amp_for

pytestmark = pytest.mark.gpu


def _need_free_hbm(gib):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()            # what the previous test's tensors occupied is only cached by torch, not in use
    free, _ = torch.cuda.mem_get_info()
    if free < gib * (1 << 30):
        pytest.skip("needs %d GiB of free HBM, %d available" % (gib, free >> 30))


def test_configs3_mode16_one_million_frames_multipath_round_trip():
    """1,048,576 mode-16 (32QAM, LDPC 14/16, zero-forcing) frames through the static 2-path channel at 30 dB: every frame
    that reports message_decoded carries exactly the payload that was sent, and ~98 % do."""
    import torch
    from mercury_amd import RxPhy
    cfg, F = 16, 1 << 20
    _need_free_hbm(60)
    rx = RxPhy(cfg, max_batch=F, agc=0, variance_source=0)           # the baseband_test variant the ZF modes are run in
    dev = torch.device("cuda:0")
    bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)          # 41 GB
    sent = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
    got = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
    stats = torch.empty((F, 6), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    rx.txgen_dev(SEED, 0, F, noise_amp_for(30.0), bb.data_ptr(), sent.data_ptr(), channel=1, stream=s)
    rx.receive_dev(bb.data_ptr(), F, got.data_ptr(), stats.data_ptr(), stream=s)
    torch.cuda.synchronize()
    decoded = stats[:, 3] == 1
    frac = float(decoded.float().mean().item())
    assert 0.97 < frac <= 1.0, frac
    assert torch.equal(got[decoded], sent[decoded])                                       # round trip on ~1M frames
    assert bool(((stats[:, 1] == 0) | ~decoded).all())                                    # decoded => CRC self-check 0
    assert int(stats[:, 0].max().item()) <= 51 and int(stats[:, 0].min().item()) >= 0
    # frames are independent: decoding the second half alone gives the same bytes (sharding invariance)
    got2 = torch.empty((F // 2, rx.payload_stride), dtype=torch.uint8, device=dev)
    stats2 = torch.empty((F // 2, 6), dtype=torch.int32, device=dev)
    rx.receive_dev(bb[F // 2:].data_ptr(), F // 2, got2.data_ptr(), stats2.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert torch.equal(got2, got[F // 2:]) and torch.equal(stats2[:, :4], stats[F // 2:, :4])
    # and the CPU oracle on a sample of the very same frames (first / middle / last + one frame that did not decode, if any):
    # payload bytes, iteration count, CRC and all-zeros flag identical
    import oraclelib
    orc = oraclelib.Oracle(cfg, 50)
    sample = [0, F // 2, F - 1]
    failed = torch.nonzero(~decoded)
    if failed.numel():
        sample.append(int(failed[0].item()))
    for f in sample:
        ref = orc.rx(bb[f].cpu().numpy().view(np.complex128).reshape(-1), oraclelib.FLAGS_BASEBAND_TEST)
        assert np.array_equal(got[f].cpu().numpy(), ref["bytes"].astype(np.uint8)), (cfg, f)
        st = stats[f].cpu().numpy()
        assert (int(st[0]), int(st[1]), int(st[2])) == (ref["iterations"], ref["crc"], ref["all_zeros"]), (cfg, f)
    del bb
    rx.close()


@pytest.mark.parametrize("iters", [5, 20, 50])
def test_configs4_ldpc_soak_share_of_one_gpu(iters):
    """BASELINE.json configs[4]: 12.5 M rate-8/16 codewords (one GPU's eighth of the 10^8 soak) of noise-only LLRs at max 5, 20
    and 50 iterations: no codeword may converge, every one must report max_iters + 1, and a second pass over the same buffer
    must reproduce every output bit (checksum of checksums)."""
    import torch
    from mercury_amd import RxPhy
    cfg, F = 6, 12_500_000
    _need_free_hbm(110)
    rx = RxPhy(cfg, max_iters=iters, max_batch=F)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    llr = torch.empty((F, 1600), dtype=torch.float32, device=dev)                        # 80 GB
    chunk = 1 << 20
    for a in range(0, F, chunk):
        llr[a: a + chunk].normal_(0.0, 0.3, generator=g)
    its = torch.empty(F, dtype=torch.int32, device=dev)
    bits = torch.empty((F, rx.K), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    sums = []
    for _ in range(2 if iters < 50 else 1):       # the 50-iteration pass is 25 s of GPU time: its determinism is covered at 5 and 20
        its.zero_()
        bits.zero_()
        rx.ldpc_decode_dev(llr.data_ptr(), F, bits.data_ptr(), its.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert int(its.min().item()) == iters + 1 and int(its.max().item()) == iters + 1
        sums.append((int(bits.to(torch.int64).sum().item()), int((bits.view(torch.int64) if rx.K % 8 == 0 else bits.to(torch.int64)).sum().item())))
    assert sums[0] == sums[-1]
    assert 0.45 < sums[0][0] / (F * rx.K) < 0.55                                          # hard decisions of noise: about half ones
    # the CPU oracle's decoder on the first / middle / last codeword of the very same buffer: hard bits and iteration count identical
    import oraclelib
    orc = oraclelib.Oracle(cfg, iters)
    for f in (0, F // 2, F - 1):
        ref_bits, ref_it = orc.ldpc_decode(llr[f].cpu().numpy())
        assert int(its[f].item()) == ref_it, (iters, f)
        assert np.array_equal(bits[f].cpu().numpy(), ref_bits.astype(np.uint8)), (iters, f)
    del llr
    rx.close()


def test_audio_loopback_transmit_byte_to_receive_byte_1024_windows():
    """The whole stack at scale, audio in the middle: 1024 messages -> transmit_byte on the GPU (filtered passband) -> capture
    windows at random delays with receiver noise (torch, on the device) -> receive_byte on the GPU. Every window must come back
    decoded with the message that went in, and the transmitter must be deterministic."""
    import torch
    from mercury_amd import RxPhy
    cfg, W = 8, 1024
    carrier = 48000.0 * 50.0 / 256 / 4 / 2 + 300
    rx = RxPhy(cfg, max_batch=W)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    msgs = torch.randint(0, 256, (W, rx.payload_bytes), dtype=torch.uint8, device=dev, generator=g)
    total, n = rx.transmit_frame_samples(), rx.receive_buffer_samples()
    audio = torch.empty((W, total), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()                                  # the library works on its own stream
    rx.transmit_byte_dev(msgs.data_ptr(), rx.payload_bytes, W, audio.data_ptr(), carrier)
    again = torch.empty_like(audio)
    rx.transmit_byte_dev(msgs.data_ptr(), rx.payload_bytes, W, again.data_ptr(), carrier)
    assert torch.equal(audio, again)
    wins = torch.randn((W, n), dtype=torch.float64, device=dev, generator=g) * 2e-3
    sym = rx.Nofdm * 4
    delays = torch.randint(5 * sym, n - total - 5 * sym, (W,), device=dev, generator=g)
    idx = delays[:, None] + torch.arange(total, device=dev)[None, :]
    wins.scatter_add_(1, idx, 2.0 * audio)                   # receiver audio gain 2 (see tests/test_transmit_byte.py)
    torch.cuda.synchronize()
    r = rx.receive_byte_dev(wins.data_ptr(), W, carrier)       # the windows stay in HBM
    assert int(r["stats"]["message_decoded"].sum()) == W
    r_host = rx.receive_byte(wins[:64].cpu().numpy(), carrier)  # same windows from host memory: same answers
    assert np.array_equal(r_host["payload"], r["payload"][:64]) and np.array_equal(r_host["stats"], r["stats"][:64])
    # 640 and all 1024 windows from host memory: the call is cut into sub-batches (320 + 320, 512 + 512) that a helper thread uploads
    # while the previous one is being received — same answers, window for window
    for n_host in (640, W):
        r_host = rx.receive_byte(wins[:n_host].cpu().numpy(), carrier)
        assert np.array_equal(r_host["payload"], r["payload"][:n_host]) and np.array_equal(r_host["stats"], r["stats"][:n_host]), n_host
    assert np.array_equal(r["payload"][:, : rx.payload_bytes], msgs.cpu().numpy())
    assert np.abs(r["stats"]["delay"] - delays.cpu().numpy()).max() <= 8 * 4          # within the guard interval's reach
    rx.close()


@pytest.mark.parametrize("cfg", list(range(17)) + [100, 101, 102])
def test_configs2_64k_frame_batches_every_mode(cfg):
    """BASELINE.json configs[2]: every mode (17 OFDM modes with the LDPC rate the reference pairs them with, 3 MFSK modes) in a
    65,536-frame batch at its operating point. Size-independent properties on the whole batch (decoded => the payload that was
    sent and CRC 0; >= 99 % decode; iteration counts in range) plus the CPU oracle on a sample of the very same frames
    (first / middle / last: payload, iteration count, CRC and all-zeros flag identical)."""
    import torch
    import oraclelib
    from conftest import OPERATING_ESN0
    from mercury_amd import RxPhy
    F = 65536
    agc, vs, flags = (0, 0, oraclelib.FLAGS_BASEBAND_TEST) if cfg in (15, 16) else (1, 1, oraclelib.FLAGS_RECEIVE_BYTE)
    rx = RxPhy(cfg, max_batch=F, agc=agc, variance_source=vs)
    _need_free_hbm(int(F * rx.frame_samples * 16 / (1 << 30)) + 8)
    dev = torch.device("cuda:0")
    bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
    sent = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
    got = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
    stats = torch.empty((F, 6), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    rx.txgen_dev(SEED, 1 << 36, F, noise_amp_for(OPERATING_ESN0[cfg] + 1.0), bb.data_ptr(), sent.data_ptr(), stream=s)
    rx.receive_dev(bb.data_ptr(), F, got.data_ptr(), stats.data_ptr(), stream=s)
    torch.cuda.synchronize()
    decoded = stats[:, 3] == 1
    assert float(decoded.float().mean().item()) >= 0.99, (cfg, float(decoded.float().mean().item()))
    nb = rx.payload_bytes
    assert torch.equal(got[decoded][:, :nb], sent[decoded][:, :nb])
    assert bool(((stats[:, 1] == 0) | ~decoded).all())
    assert int(stats[:, 0].max().item()) <= 51 and int(stats[:, 0].min().item()) >= 0
    orc = oraclelib.Oracle(cfg, 50)
    for f in (0, F // 2, F - 1):
        ref = orc.rx(bb[f].cpu().numpy().view(np.complex128).reshape(-1), flags)
        assert np.array_equal(got[f].cpu().numpy(), ref["bytes"].astype(np.uint8)), (cfg, f)
        st = stats[f].cpu().numpy()
        assert (int(st[0]), int(st[1]), int(st[2])) == (ref["iterations"], ref["crc"], ref["all_zeros"]), (cfg, f)
    del bb
    rx.close()
