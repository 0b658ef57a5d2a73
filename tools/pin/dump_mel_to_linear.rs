// Pin kit, step 3 (runs where the `griffin-lim` crate builds: the reference pins it at
// e6415314cf3309787e54d9ff2768454373e82a5c, /root/reference/Cargo.lock:666-668).
//
// Drop this file into the reference checkout as examples/dump_mel_to_linear.rs, copy tools/pin/fixed_mel.npy next to
// Cargo.toml (python tools/pin/fixed_mel.py writes it), then
//
//     cargo run --release --example dump_mel_to_linear
//
// It calls exactly what the reference calls (src/tacotron2/mod.rs:441-458, src/lib.rs:141) and writes
//   reference_mel_basis.npy       create_mel_filter_bank(22050, 1024, 80, 0, Some(8000))            (80 x 513)
//   reference_audio.npy           GriffinLim::infer(&mel) on the fixed mel                            (one run: the phase is random)
//   reference_audio_x4.npy        the same mel + ln 4 (4 x the linear mel): if the crate normalises its output (the slide WAVs say
//                                 RMS 0.1) the two audios have the SAME RMS, otherwise 4^(1/1.7) apart
//   reference_mel_to_linear.npy   the crate's deterministic first stage, IF it is reachable: the crate keeps it private at the
//                                 pinned commit, so expose it for the dump -- in the crate's src/lib.rs make the function
//                                 `infer` calls first (the NNLS / pseudo-inverse step that turns the (80 x F) mel into the
//                                 (513 x F) magnitude, before the Griffin-Lim loop) `pub`, and call it at the marked line.
// Copy the .npy files to tests/golden/ of the MI355X build: tests/test_gpu_reference_pinned.py picks them up.
use griffin_lim::mel::create_mel_filter_bank;
use griffin_lim::GriffinLim;
use ndarray::Array2;
use ndarray_npy::{read_npy, write_npy};

fn rms(x: &[f32]) -> f64 {
    (x.iter().map(|v| (*v as f64) * (*v as f64)).sum::<f64>() / x.len() as f64).sqrt()
}

fn main() -> anyhow::Result<()> {
    let mel: Array2<f32> = read_npy("fixed_mel.npy")?; // (80, 64), natural-log mel with a -11.5 floor
    let mel_basis = create_mel_filter_bank(22050.0, 1024, 80, 0.0, Some(8000.0)); // mod.rs:453
    write_npy("reference_mel_basis.npy", &mel_basis)?;
    let vocoder = GriffinLim::new(mel_basis.clone(), 1024 - 256, 1.7, 30, 0.99)?; // mod.rs:456

    let audio = vocoder.infer(&mel)?; // lib.rs:141
    println!("audio: {} samples, rms {:.6}, peak {:.6}", audio.len(), rms(&audio), audio.iter().fold(0f32, |m, v| m.max(v.abs())));
    write_npy("reference_audio.npy", &ndarray::Array1::from(audio))?;

    let louder = mel.mapv(|v| v + 4f32.ln());
    let audio4 = vocoder.infer(&louder)?;
    println!("audio (mel x 4): rms {:.6}  (equal to the line above <=> the crate normalises its output)", rms(&audio4));
    write_npy("reference_audio_x4.npy", &ndarray::Array1::from(audio4))?;

    // ---- the deterministic first stage: uncomment once the crate's mel -> linear function is `pub` ----------------------
    // let linear: Array2<f32> = vocoder.mel_to_linear(&mel)?;      // <- the name at the pinned commit may differ
    // write_npy("reference_mel_to_linear.npy", &linear)?;           // (513 x 64) or (64 x 513): the test accepts either
    Ok(())
}
