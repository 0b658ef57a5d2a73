"""CPU-only checks of the drop-in boundary: libxdtts_hip.so loads, exports every symbol that
include/xdtts.h declares, refuses to run without a GPU (no CPU fallback), and its host-side front
end reproduces the reference's known-answer tests (src/tacotron2/mod.rs:465-508,
src/phonemes.rs:785-811)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "xdtts.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xdtts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    names = declared_symbols()
    assert len(names) >= 35
    raw = C.CDLL(pkg.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libxdtts_hip.so does not export %s" % n
    # and the Python binding lists exactly the declared set
    assert sorted(pkg.SYMBOLS) == names


def test_no_gpu_means_error_not_fallback(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.XdttsError) as e:
        pkg.Tacotron2.synthetic()
    assert e.value.status == pkg.XDTTS_ERR_NO_DEVICE
    with pytest.raises(pkg.XdttsError) as e:
        pkg.create_griffin_lim()
    assert e.value.status == pkg.XDTTS_ERR_NO_DEVICE


def test_synthesize_batch_rejects_bad_arguments_without_touching_a_device(pkg):
    """xdtts_synthesize_batch (XdTts::infer for several utterances): argument errors come back as status
    codes with a message, before anything is launched -- so this runs without a GPU too."""
    import ctypes as C
    n = C.c_size_t()
    st = pkg.lib.xdtts_synthesize_batch(None, None, None, None, 1, 1, None, 1, None, None, None, C.byref(n), None, None)
    assert st == pkg.XDTTS_ERR_BAD_ARG
    assert b"null" in pkg.lib.xdtts_last_error()


def test_default_opts_are_the_reference_constants(pkg):
    o = pkg.default_opts()
    assert abs(o.gate_threshold - 0.6) < 1e-7 and o.max_steps == 1000  # mod.rs:279-280
    assert o.max_chunk == 100  # mod.rs:363,369-371,399
    assert o.fixed_steps == 0 and o.dropout_mode == 1


def test_tensor_table_matches_oracle(pkg, orc):
    ours = pkg.tensor_table()
    theirs = orc.tensor_table()
    assert [(n, s, o) for n, s, o in ours] == [(n, s, o) for n, s, o, _ in theirs]
    assert pkg.lib.xdtts_tensor_total() == 28_200_481


def test_symbol_table(pkg):
    syms = pkg.generate_id_list()
    assert len(syms) == 148  # mod.rs:90-122
    assert syms[0] == "<PAD>" and syms[11] == " " and syms[7] == "." and syms[12] == "A" and syms[38] == "a"
    assert syms[64] == "AA" and syms[-1] == "ZH"


def test_reference_id_known_answers(pkg):
    k = KATS["correct_phoneme_id_output"]  # mod.rs:465-492
    assert [pkg.unit_id(u) for u in k["units"]] == k["expected"]
    k = KATS["correct_char_id_output"]  # mod.rs:494-508
    ids = [pkg.unit_id(c, as_character=(c not in "!")) for c in k["text"]]
    assert ids == k["expected"]


def test_unit_lookup_edge_cases(pkg):
    assert pkg.unit_id("AA3") == pkg.unit_id("AA")  # unknown stress mark: first entry of that phone (phonemes.rs:627-660)
    assert pkg.unit_id("B") == 88  # ARPAbet wins over the character 'B' (phonemes.rs:469-482)
    assert pkg.unit_id("B", as_character=True) == 13
    assert pkg.unit_id("<UNK>") is None and pkg.unit_id("%") is None  # dropped (mod.rs:403-406)
    assert pkg.unit_id("<PAD>") == 0 and pkg.unit_id(" ") == 11
    assert list(pkg.units_to_ids(["HH", "%", "AH0", " "])) == [106, 73, 11]


def test_find_splits_reference_kat(pkg):
    k = KATS["split_units"]  # phonemes.rs:785-811
    text = k["text"]
    ids = np.array([pkg.unit_id(c) if c in ".," else pkg.unit_id(c, True) for c in text])
    assert len(ids) == len(text)
    sp = list(pkg.find_splits(ids, k["max_size"]))
    assert len(sp) == 3
    assert text[sp[0]] == "." and text[sp[1]] == "." and sp[0] < sp[1]
    assert sp[1] < sp[2] < sp[1] + 11 and text[sp[2]] == " "


def find_splits_py(scores, n, max_size):
    """The chunking heuristic of src/phonemes.rs:681-753 restated independently in Python (scores =
    split_score per position) -- the cross-check for the C++ host implementation."""
    marks = [(i, s) for i, s in enumerate(scores) if s > 0]
    results = [0] + [i for i, s in marks if s > 2]
    threshold, scan, fresh = 1, True, []
    while scan:
        scan = False
        last_ref = n
        for index in reversed(results):
            if last_ref - index > max_size:
                scan = True
                fresh += [i for i, s in marks if i < last_ref and i > index + 1 and s > threshold]
            last_ref = index
        if scan:
            results = sorted(results + fresh)
            fresh = []
        if threshold > 0:
            threshold -= 1
        else:
            scan = False
    merged, running, last_insert = [], 0, 0
    for i in results:
        if (i - last_insert) + running > max_size:
            merged.append(last_insert)
            running = i - last_insert
        else:
            running += i - last_insert
        last_insert = i
    if running + (n - last_insert) > max_size and results:
        merged.append(results[-1])
    out = []
    for m in merged:  # Vec::dedup
        if not out or out[-1] != m:
            out.append(m)
    return out


def test_find_splits_matches_independent_restatement(pkg):
    rng = np.random.default_rng(0)
    for trial in range(60):
        n = int(rng.integers(1, 400))
        window = int(rng.choice([10, 25, 100]))
        ids = 64 + rng.integers(0, 84, size=n)
        ids[rng.random(n) < 0.18] = 11
        ids[rng.random(n) < 0.04] = 7
        ids[rng.random(n) < 0.03] = 6
        ids[rng.random(n) < 0.01] = 10
        scores = [pkg.lib.xdtts_split_score(int(i)) for i in ids]
        sp = list(pkg.find_splits(ids, window))
        assert sp == find_splits_py(scores, n, window), (trial, n, window)
        assert all(0 <= s <= n for s in sp) and sp == sorted(sp)
        if n <= window:
            assert sp == []
    assert [pkg.lib.xdtts_split_score(i) for i in (7, 10, 2, 0, 6, 9, 11, 8, 1, 64)] == [3, 3, 3, 3, 2, 2, 1, 0, 0, 0]  # phonemes.rs:663-671


def test_bench_utterance_chunks(pkg):
    from conftest import synth_ids

    ids = synth_ids(120)
    assert list(pkg.find_splits(ids, 100)) == [95]  # 95 + 25 ids


def test_mel_filter_bank_is_host_side_and_bit_exact(pkg, orc):
    B = pkg.create_mel_filter_bank(22050.0, 1024, 80, 0.0, 8000.0)
    assert np.array_equal(B, orc.mel_filter_bank(22050.0, 1024, 80, 0.0, 8000.0))
    # fmax = None -> sr / 2
    assert np.array_equal(pkg.create_mel_filter_bank(16000.0, 512, 40, 0.0, None), orc.mel_filter_bank(16000.0, 512, 40, 0.0, 8000.0))


def _build_host_smoke(pkg, tmp_path):
    import subprocess

    exe = str(tmp_path / "host_smoke")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(
        ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_smoke.cpp"), "-o", exe,
         "-L", libdir, "-lxdtts_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    )
    return subprocess.check_output([exe]).decode().splitlines()


def test_cpp_host_mirror_links_with_plain_gxx(pkg, tmp_path):
    """include/xdtts_host.hpp (the C++ mirror of Tacotron2::load/infer, GriffinLim::new/infer) builds
    with plain g++ against the C ABI -- no HIP headers, no torch."""
    lines = _build_host_smoke(pkg, tmp_path)
    assert lines[0].split() == [str(x) for x in KATS["correct_char_id_output"]["expected"]]
    if pkg.device_count() == 0:
        assert lines[1] == "no device: status %d" % pkg.XDTTS_ERR_NO_DEVICE


@pytest.mark.gpu
def test_cpp_host_mirror_sanity_on_gpu(pkg, tmp_path):
    """The reference's tacotron_sanity_test (mod.rs:511-522) through the C++ mirror: 80 rows, >0 cols."""
    lines = _build_host_smoke(pkg, tmp_path)
    rows, _x, cols = lines[1].split()[1:]
    assert int(rows) == 80 and int(cols) > 0
    assert int(lines[2].split()[1]) == 256 * (int(cols) - 1)
    many = lines[3].split()
    assert many[:2] == ["many", "2:"] and len(many) == 4
    for pair in many[2:]:   # (frames, samples) per utterance: max_steps = 5 frames -> 256 * 4 samples
        f, n = (int(x) for x in pair.split("/"))
        assert f > 0 and n == 256 * (f - 1)


# ---- output stage (src/lib.rs:25-30,125-176): host-side, runs without a GPU ------------------------

def test_audio_to_i16_follows_rust_float_to_int_cast(pkg):
    """`(sample * i16::MAX as f32) as i16`: truncation toward zero, saturation, NaN -> 0."""
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 0.999985, 1.5, -1.5, 3.0e-5, -3.0e-5, 6.2e-5, np.nan, np.inf, -np.inf], np.float32)
    got = pkg.audio_to_i16(x)
    want = []
    for v in x:
        s = np.float32(v) * np.float32(32767.0)
        if np.isnan(s):
            want.append(0)
        else:
            want.append(int(max(-32768.0, min(32767.0, np.trunc(s)))))
    assert got.dtype == np.int16 and got.tolist() == want
    assert got[:5].tolist() == [0, 32767, -32767, 16383, -16383]  # truncation, not rounding


def test_silence_length_of_a_break(pkg):
    assert pkg.silence_samples(0.0) == 0
    assert pkg.silence_samples(1.0) == 22050
    assert pkg.silence_samples(0.25) == 5513   # 5512.5 rounds away from zero (f32::round)
    assert pkg.silence_samples(0.00002) == 0
    assert pkg.silence_samples(0.5, 16000) == 8000


def test_silence_length_follows_the_f32_arithmetic_of_write_silence(pkg):
    """src/lib.rs:166: `(sample_rate as f32 * duration.as_secs_f32()).round() as u32` -- every step in
    f32 (Duration::as_secs_f32 = secs as f32 + nanos as f32 / 1e9).  Restated with numpy float32."""
    f32 = np.float32
    rng = np.random.Generator(np.random.PCG64(11))
    cases = [(0, 0), (1, 0), (0, 250_000_000), (0, 20_000), (3, 999_999_999), (0, 22_675_737), (0, 68_027)]
    cases += [(int(s), int(n)) for s, n in zip(rng.integers(0, 40, 300), rng.integers(0, 1_000_000_000, 300))]
    # durations that land on a .5 sample boundary in exact arithmetic (odd multiples of 1/44100 s)
    cases += [(k // 44100, int(round((k % 44100) / 44100 * 1e9))) for k in range(1, 4001, 2)]
    for secs, nanos in cases:
        dur = f32(f32(secs) + f32(f32(nanos) / f32(1e9)))
        prod = f32(f32(22050) * dur)
        want = int(np.floor(np.abs(prod) + f32(0.5)))          # f32::round: half away from zero
        assert pkg.silence_samples_duration(secs, nanos) == want, (secs, nanos)
        assert pkg.silence_samples(float(dur)) == want, (secs, nanos)


def test_wav_file_has_the_reference_spec(pkg, tmp_path):
    import wave

    rng = np.random.Generator(np.random.PCG64(5))
    audio = rng.uniform(-1.2, 1.2, size=4321).astype(np.float32)
    path = str(tmp_path / "out.wav")
    pkg.write_wav(path, audio)
    with wave.open(path, "rb") as w:
        assert (w.getnchannels(), w.getframerate(), w.getsampwidth(), w.getnframes()) == (1, 22050, 2, 4321)
        pcm = np.frombuffer(w.readframes(4321), dtype="<i2")
    assert np.array_equal(pcm, pkg.audio_to_i16(audio))
    assert os.path.getsize(path) == 44 + 2 * 4321
    pkg.write_wav(str(tmp_path / "empty.wav"), np.zeros(0, np.float32))
    assert os.path.getsize(str(tmp_path / "empty.wav")) == 44
    with pytest.raises(pkg.XdttsError):
        pkg.write_wav(str(tmp_path / "no_such_dir" / "x.wav"), audio)


def test_mel_npy_dump_is_readable_by_numpy(pkg, tmp_path):
    mel = np.arange(80 * 37, dtype=np.float32).reshape(80, 37) / 7.0
    path = str(tmp_path / "mel.npy")
    pkg.write_mel_npy(path, mel)
    back = np.load(path)
    assert back.dtype == np.float32 and back.shape == (80, 37) and np.array_equal(back, mel)
    assert not np.isfortran(back)


def test_real_time_factor_formula(pkg):
    assert pkg.real_time_factor(1.0, 22050) == pytest.approx(1.0)
    assert pkg.real_time_factor(0.0106, 204544) == pytest.approx(0.0106 / (204544 / 22050.0))
    assert pkg.real_time_factor(1.0, 0) == 0.0


def test_the_loaded_library_was_built_from_the_sources_in_the_tree(pkg):
    """The .so files are git-ignored and travel prebuilt: xdtts_build_info() carries the sha256 the Makefile took over
    csrc/* and include/xdtts.h, and it must equal the hash of the sources next to the library (a stale .so fails here)."""
    info = pkg.build_info()
    assert info["arch"] == "gfx950"
    assert info["src_sha256"] == pkg.source_hash(), "libxdtts_hip.so is stale: run `make -C xd-tts_amd` (or __graft_entry__.build())"


def test_default_device_follows_the_environment(pkg):
    """XDTTS_DEVICE_DEFAULT (-1) as a device_id = environment variable XDTTS_DEVICE, read at every handle creation (how a host that
    keeps the reference's device-less constructors, src/lib.rs:40-58, is spread over a node's GPUs); xdtts_default_device says what it
    resolves to.  Host side only: no device is touched."""
    old = os.environ.pop("XDTTS_DEVICE", None)
    try:
        assert pkg.DEVICE_DEFAULT == -1 and pkg.default_device() == 0
        os.environ["XDTTS_DEVICE"] = "5"
        assert pkg.default_device() == 5
        for bad in ("gpu1", "-2", "3x", "99999"):
            os.environ["XDTTS_DEVICE"] = bad
            with pytest.raises(pkg.XdttsError) as e:
                pkg.default_device()
            assert "XDTTS_DEVICE" in str(e.value)
    finally:
        os.environ.pop("XDTTS_DEVICE", None)
        if old is not None:
            os.environ["XDTTS_DEVICE"] = old
