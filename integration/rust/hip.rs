//! src/hip/mod.rs of the reference crate under `--features hip` — the COARSE shim: the three public calls the bench harness
//! makes (`/root/reference/src/benches/bench.rs:54-66`) routed to `liblasso_prover.so` (include/lasso_prover.h), results brought
//! back as the crate's own types through `deserialize_compressed`.
//!
//!   DensifiedRepresentation::from_lookup_indices(&nz, log_m)        densified.rs:22    -> HipDensified::from_lookup_indices
//!   dense.commit::<G>(&gens)                                        densified.rs:78    -> HipDensified::commit
//!   SparsePolynomialEvaluationProof::<G,C,M,S>::prove(..)           surge.rs:119-125   -> prove_hip::<G,C,M,S>
//!
//! Nothing of the protocol is re-written in Rust on this path: the transcript schedule and the O(log n) scalar work run inside
//! liblasso_prover.so (lasso_amd/host/prover.hpp — the C++ mirror of the reference's host), every O(n) loop inside liblasso_hip.so.
//! The Merlin transcript and the random tape are the CALLER'S: `prove_hip` takes `&mut Transcript` and `&mut RandomTape<G>` exactly as
//! surge.rs:119-125 does and hands them to the library as two callbacks each (lasso_transcript_vtbl), so a transcript that already holds
//! state binds the proof to that state — byte for byte what the CPU prover produces from the same transcript.  What crosses the boundary: the lookup indices (`Vec<[usize; C]>` is n x C u64, row-major, passed as is), the point
//! `r` (ark-ff's in-memory Montgomery limbs, passed as is), and ark-serialize's compressed wire bytes on the way back.
//! The finely-grained ABI (include/lasso_hip.h, ffi.rs first block) stays available to a maintainer who wants the protocol in Rust and
//! only the loops on the device; INTEGRATION.md §3 maps each loop to its entry point.
//!
//! NOT COMPILED HERE: the build image has no cargo/rustc.  Written against ark-* 0.4.2, merlin 3.0 as pinned in the
//! reference's Cargo.toml; the layout asserts below are what makes passing `&[F]` as `*const lasso_fr` sound.
#![cfg(feature = "hip")]

use std::cell::RefCell;
use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::c_void;

use merlin::Transcript;

use ark_ec::CurveGroup;
use ark_ff::PrimeField;
use ark_serialize::CanonicalDeserialize;

use crate::hip::ffi::*;
use crate::lasso::surge::{SparsePolyCommitmentGens, SparsePolynomialCommitment, SparsePolynomialEvaluationProof};
use crate::poly::dense_mlpoly::PolyCommitmentGens;
use crate::poly::dense_mlpoly::PolyCommitment;
use crate::utils::random::RandomTape;
use crate::subtables::{
  and::AndSubtableStrategy, lt::LTSubtableStrategy, or::OrSubtableStrategy, range_check::RangeCheckSubtableStrategy,
  xor::XorSubtableStrategy, SubtableStrategy,
};

pub mod ffi; // integration/rust/ffi.rs (generated from the two C headers)

// ---- layout guards: the ABI passes ark-ff / ark-ec memory through unchanged -------------------------------------------------------
const _: () = assert!(std::mem::size_of::<ark_curve25519::Fr>() == 32 && std::mem::align_of::<ark_curve25519::Fr>() == 8);
const _: () = assert!(std::mem::size_of::<lasso_fr>() == 32 && std::mem::size_of::<lasso_affine>() == 64 && std::mem::size_of::<lasso_point>() == 128);
const _: () = assert!(std::mem::size_of::<ark_curve25519::EdwardsAffine>() == 64); // {x, y}; no infinity flag on TE curves
const _: () = assert!(std::mem::size_of::<ark_curve25519::EdwardsProjective>() == 128); // {x, y, t, z}
const _: () = assert!(std::mem::size_of::<usize>() == 8); // Vec<[usize; C]> is uploaded as n x C u64

fn last_error() -> String {
  unsafe { CStr::from_ptr(lasso_host_last_error()).to_string_lossy().into_owned() }
}
/// The reference prover panics on contract violations (surge.rs:131, memory_checking.rs:689); a non-zero status becomes the same.
fn chk(rc: i32, what: &str) {
  assert!(rc == 0, "{what} failed ({rc}): {}", last_error());
}

/// `SubtableStrategy` -> the runtime descriptor the library takes (include/lasso_hip.h `lasso_strategy`).
pub trait HipStrategy<F: PrimeField, const C: usize, const M: usize>: SubtableStrategy<F, C, M> {
  const KIND: i32;
  const LOG_R: u32 = 0;
  fn descriptor() -> lasso_strategy {
    lasso_strategy { kind: Self::KIND, c: C as u32, log_m: M.trailing_zeros(), log_r: Self::LOG_R }
  }
}
impl<F: PrimeField, const C: usize, const M: usize> HipStrategy<F, C, M> for AndSubtableStrategy { const KIND: i32 = LASSO_AND; }
impl<F: PrimeField, const C: usize, const M: usize> HipStrategy<F, C, M> for OrSubtableStrategy { const KIND: i32 = LASSO_OR; }
impl<F: PrimeField, const C: usize, const M: usize> HipStrategy<F, C, M> for XorSubtableStrategy { const KIND: i32 = LASSO_XOR; }
impl<F: PrimeField, const C: usize, const M: usize> HipStrategy<F, C, M> for LTSubtableStrategy { const KIND: i32 = LASSO_LT; }
impl<F: PrimeField, const C: usize, const M: usize, const LOG_R_: usize> HipStrategy<F, C, M> for RangeCheckSubtableStrategy<LOG_R_> {
  const KIND: i32 = LASSO_RANGE;
  const LOG_R: u32 = LOG_R_ as u32;
}

/// One device context + host prover (`lasso_host`).  One per prover thread; the library is not re-entrant on a context.
pub struct HipProver {
  h: *mut lasso_host,
  /// device-resident generator tables, one entry per distinct `SparsePolyCommitmentGens` the caller has passed (see `gens_on_device`)
  gens_cache: RefCell<Vec<(GensKey, *mut lasso_host_gens)>>,
}
impl HipProver {
  pub fn new(device: i32) -> Self {
    let mut h = std::ptr::null_mut();
    chk(unsafe { lasso_host_create(device, &mut h) }, "lasso_host_create");
    let p = HipProver { h, gens_cache: RefCell::new(Vec::new()) };
    p.self_test();
    p
  }

  /// Start-up self-test of the layout assumption the whole boundary rests on: ark-ff's in-memory `Fr` IS the library's `lasso_fr`.
  /// `F::from(7)` and `F::from(6)` go to the device as raw bytes, the device lifts the integers 7 and 6 itself (`lasso_fr_from_u32`), and both pairs must come
  /// back as the same 32 bytes — which also read back as 7 and 6 through ark-ff.  A build of ark-ff with another limb order, radix or Montgomery constant fails here,
  /// at start-up, instead of producing proofs nobody can verify.
  fn self_test(&self) {
    // the scalar field of the library pair this binary is linked against (the BN254 pair must be checked against ark_bn254::Fr, not curve25519's)
    #[cfg(feature = "hip-bn254")]
    type F = ark_bn254::Fr;
    #[cfg(not(feature = "hip-bn254"))]
    type F = ark_curve25519::Fr;
    use ark_ff::PrimeField as _;
    let ctx = unsafe { lasso_host_ctx(self.h) };
    let host_side = [F::from(7u64), F::from(6u64)];
    let ints: [u32; 2] = [7, 6];
    let (mut d_ints, mut d_fr) = (std::ptr::null_mut::<c_void>(), std::ptr::null_mut::<c_void>());
    unsafe {
      chk(lasso_alloc(ctx, 8, &mut d_ints), "lasso_alloc");
      chk(lasso_alloc(ctx, 64, &mut d_fr), "lasso_alloc");
      chk(lasso_upload(ctx, d_ints, ints.as_ptr() as *const c_void, 8), "lasso_upload");
      chk(lasso_fr_from_u32(ctx, d_ints as *const u32, 2, d_fr as *mut lasso_fr), "lasso_fr_from_u32");
      let mut back = [F::from(0u64); 2];
      chk(lasso_download(ctx, back.as_mut_ptr() as *mut c_void, d_fr, 64), "lasso_download");
      assert!(back == host_side, "ark-ff's Fr memory form is not the loaded library's field-element layout (4 x u64 LE, Montgomery R = 2^256) — or the library pair is the other curve's");
      assert!(back[0].into_bigint().0 == [7, 0, 0, 0]);
      // and the other direction: the host's bytes, through the device and back, unchanged
      chk(lasso_upload(ctx, d_fr, host_side.as_ptr() as *const c_void, 64), "lasso_upload");
      chk(lasso_download(ctx, back.as_mut_ptr() as *mut c_void, d_fr, 64), "lasso_download");
      assert!(back == host_side);
      chk(lasso_free(ctx, d_ints), "lasso_free");
      chk(lasso_free(ctx, d_fr), "lasso_free");
    }
  }
}

// ---- merlin::Transcript behind include/lasso_prover.h's two callbacks ------------------------------------------------------------
// `user` is the caller's `&mut Transcript`.  merlin's signatures ask for `&'static [u8]` labels: the library's labels are string literals of the loaded
// shared object (include/lasso_prover.h states the lifetime), so extending the slice's lifetime is sound for as long as the library stays loaded.
unsafe extern "C" fn merlin_append(user: *mut c_void, label: *const u8, label_len: usize, msg: *const u8, msg_len: usize) {
  let t = &mut *(user as *mut Transcript);
  let label: &'static [u8] = std::slice::from_raw_parts(label, label_len);
  t.append_message(label, std::slice::from_raw_parts(msg, msg_len));
}
unsafe extern "C" fn merlin_challenge(user: *mut c_void, label: *const u8, label_len: usize, dest: *mut u8, dest_len: usize) {
  let t = &mut *(user as *mut Transcript);
  let label: &'static [u8] = std::slice::from_raw_parts(label, label_len);
  t.challenge_bytes(label, std::slice::from_raw_parts_mut(dest, dest_len));
}
const MERLIN_VTBL: lasso_transcript_vtbl = lasso_transcript_vtbl { append_message: Some(merlin_append), challenge_bytes: Some(merlin_challenge) };
// RandomTape<G> { tape: Transcript, .. } (utils/random.rs:9-12): the shim lives inside the crate, `tape` gets `pub(crate)`.
// ---- the caller's generators --------------------------------------------------------------------------------------------------------
// `commit` and `prove` take `gens: &SparsePolyCommitmentGens<G>` (densified.rs:78-81, surge.rs:119-125) and so do the shims below: the POINTS the caller holds are
// uploaded (lasso_host_gens_from_points) — the library derives nothing, so no restatement of arkworks' SHAKE256 -> ChaCha20Rng -> G::rand stream is in the
// prover's trust base.  Building the device tables costs tens of milliseconds, so they are cached per HipProver under a key taken from the points themselves.
/// Cache key (the first form normalised and copied every point of all three sets on EVERY commit / prove call just to form the key — tens of thousands of
/// points at 2^24, inside the timed `SparsePoly.prove` span): what identifies a generator object cheaply is where its three `G` vectors live and how long they are, the raw
/// bytes of a few PROJECTIVE points of each set (first, middle, last, and `h`: no normalisation, 4 x 128 bytes per set), and the shape the tables were built for.
/// Two different generator sets that agree on all of it do not occur; a `Vec` that was dropped and re-allocated at the same address with other points differs in the probe.
#[derive(PartialEq, Eq, Clone)]
struct GensKey { addrs: [usize; 3], sizes: [usize; 3], shape: [usize; 4], probe: Vec<u8> }

fn key_of<G: CurveGroup>(gens: &SparsePolyCommitmentGens<G>, shape: [usize; 4]) -> GensKey {
  let sets = [&gens.gens_combined_l_variate, &gens.gens_combined_log_m_variate, &gens.gens_derefs];
  let mut probe = Vec::with_capacity(3 * 4 * std::mem::size_of::<G>());
  let raw = |p: &G| unsafe { std::slice::from_raw_parts(p as *const G as *const u8, std::mem::size_of::<G>()) }.to_vec();
  for s in sets.iter() {
    let v = &s.gens.gens_n.G;
    for &i in &[0, v.len() / 2, v.len() - 1] { probe.extend_from_slice(&raw(&v[i])); }
    probe.extend_from_slice(&raw(&s.gens.gens_n.h));
  }
  GensKey {
    addrs: [sets[0].gens.gens_n.G.as_ptr() as usize, sets[1].gens.gens_n.G.as_ptr() as usize, sets[2].gens.gens_n.G.as_ptr() as usize],
    sizes: [sets[0].gens.gens_n.G.len(), sets[1].gens.gens_n.G.len(), sets[2].gens.gens_n.G.len()],
    shape, probe,
  }
}

/// [gens_n.G[0..n), gens_1.G[0], gens_n.h] as affine points in ark-ec's memory form (= lasso_affine: x then y, Montgomery limbs), one batch inversion
/// (CurveGroup::normalize_batch, as commitments.rs:87 does before every MSM).  Only on a cache miss.
fn flatten<G: CurveGroup>(g: &PolyCommitmentGens<G>) -> Vec<G::Affine> {
  let d = &g.gens; // DotProductProofGens { gens_n, gens_1 } (dot_product.rs:139-150)
  assert!(d.gens_1.G.len() == 1 && d.gens_1.h == d.gens_n.h, "gens_1 / gens_n must share h (split_at, commitments.rs:54-71)");
  let mut pts: Vec<G> = d.gens_n.G.clone();
  pts.push(d.gens_1.G[0]);
  pts.push(d.gens_n.h);
  G::normalize_batch(&pts)
}

/// at most this many generator objects stay resident per HipProver (each holds its tables in HBM: ~0.5 MB per generator with the byte multiples); the least recently
/// used one is freed when a new one arrives.  `HipProver::drop_gens` frees one explicitly.
const GENS_CACHE_MAX: usize = 4;

impl HipProver {
  /// the device-side twin of `gens`, built on first sight.  c, s, log_m: the shape (surge.rs:39-47); num_memories = 0 where the strategy is not known (commit).
  fn gens_on_device<G: CurveGroup>(&self, gens: &SparsePolyCommitmentGens<G>, c: usize, s: usize, num_memories: usize, log_m: usize) -> *mut lasso_host_gens {
    // `commit` (num_memories = 0: the strategy is not known there) is served by any object built for the same c, s, log_m — prepare_gens / prove build it with the real count
    let key = key_of(gens, [c, s, num_memories, log_m]);
    {
      let mut cache = self.gens_cache.borrow_mut();
      let hit = cache.iter().position(|(k, _)| *k == key || (num_memories == 0 && k.addrs == key.addrs && k.sizes == key.sizes && k.probe == key.probe && k.shape[0] == c && k.shape[1] == s && k.shape[3] == log_m));
      if let Some(i) = hit { let e = cache.remove(i); let g = e.1; cache.push(e); return g; }   // most recently used last
    }
    // miss: normalise, repack (TE: {x, y}; SW: {x, y, infinity} is 72 bytes and is repacked by affine_to_abi), upload, build the tables
    let sets = [flatten(&gens.gens_combined_l_variate), flatten(&gens.gens_combined_log_m_variate), flatten(&gens.gens_derefs)];
    let raw: Vec<Vec<lasso_affine>> = sets.iter().map(|v| v.iter().map(affine_to_abi::<G>).collect()).collect();
    let mut g = std::ptr::null_mut();
    chk(unsafe { lasso_host_gens_from_points(self.h, c, s, num_memories, log_m, raw[0].as_ptr(), raw[0].len(), raw[1].as_ptr(), raw[1].len(), raw[2].as_ptr(), raw[2].len(), &mut g) },
        "lasso_host_gens_from_points");
    let mut cache = self.gens_cache.borrow_mut();
    if cache.len() >= GENS_CACHE_MAX { let (_, old) = cache.remove(0); unsafe { lasso_host_gens_free(old) } }
    cache.push((key, g));
    g
  }

  /// Upload `gens` and build EVERY device table a proof over it reads — window, digit-multiple and byte-multiple tables (lasso_host_gens_prepare) — now, outside any timed span.
  /// The reference builds its generators outside the spans it times (bench.rs:56 sits between `Densify` and `DensifiedRepresentation.commit`); without this call the first
  /// `commit` pays normalize_batch + upload + k_precompute_table / _multiples / _tab8 (tens of ms) inside its span, and bench.py's `commit_warm_s` / `ms_per_step` would not line up
  /// with the Rust harness's spans (INTEGRATION.md §2 "which span pays what").
  pub fn prepare_gens<G: CurveGroup>(&self, gens: &SparsePolyCommitmentGens<G>, c: usize, s: usize, num_memories: usize, log_m: usize) {
    let g = self.gens_on_device(gens, c, s, num_memories, log_m);
    chk(unsafe { lasso_host_gens_prepare(g) }, "lasso_host_gens_prepare");
  }

  /// free the device tables held for `gens` (any shape); they are rebuilt on next sight
  pub fn drop_gens<G: CurveGroup>(&self, gens: &SparsePolyCommitmentGens<G>) {
    let probe = key_of(gens, [0; 4]);
    self.gens_cache.borrow_mut().retain(|(k, g)| { let same = k.addrs == probe.addrs && k.sizes == probe.sizes && k.probe == probe.probe; if same { unsafe { lasso_host_gens_free(*g) } } !same });
  }
}
/// ark-ec affine point -> `lasso_affine`: x, y as ark-ff's Montgomery limbs; generators are never the point at infinity
fn affine_to_abi<G: CurveGroup>(p: &G::Affine) -> lasso_affine {
  use ark_ec::AffineRepr;
  let (x, y) = p.xy().expect("a generator at infinity");
  let mut out = lasso_affine { x: [0; 4], y: [0; 4] };
  // BaseField = Fp256<MontBackend<_, 4>>: `.0` is the BigInt of Montgomery limbs — copied as they lie in memory
  assert!(std::mem::size_of_val(x) == 32 && std::mem::size_of_val(y) == 32);
  unsafe {
    std::ptr::copy_nonoverlapping(x as *const _ as *const u64, out.x.as_mut_ptr(), 4);
    std::ptr::copy_nonoverlapping(y as *const _ as *const u64, out.y.as_mut_ptr(), 4);
  }
  out
}
impl Drop for HipProver {
  fn drop(&mut self) {
    for (_, g) in self.gens_cache.borrow_mut().drain(..) { unsafe { lasso_host_gens_free(g) } }
    unsafe { lasso_host_destroy(self.h) }
  }
}

/// `DensifiedRepresentation<F, C>` with dim / read / final resident in HBM (densified.rs:8-20: the `pub` fields are only read inside the crate).
pub struct HipDensified<'a, F: PrimeField, const C: usize> {
  d: *mut lasso_host_dense,
  pub s: usize,
  pub log_m: usize,
  pub m: usize,
  p: &'a HipProver,
  _p: PhantomData<F>,
}
impl<'a, F: PrimeField, const C: usize> HipDensified<'a, F, C> {
  /// densified.rs:22-75.  `indices` is passed as the reference holds it.  Panics if an index is >= 2^log_m (the reference indexes out of bounds there).
  pub fn from_lookup_indices(p: &'a HipProver, indices: &Vec<[usize; C]>, log_m: usize) -> Self {
    let mut d = std::ptr::null_mut();
    chk(unsafe { lasso_host_densify(p.h, indices.as_ptr() as *const u64, indices.len(), C, log_m, &mut d) }, "lasso_host_densify");
    HipDensified { d, s: indices.len().next_power_of_two(), log_m, m: 1 << log_m, p, _p: PhantomData }
  }

  /// densified.rs:78-96.  The library returns `[u64 L1][L1 x 32 B][u64 L2][L2 x 32 B]` = the two `PolyCommitment { C: Vec<G> }` in
  /// ark-serialize's compressed form, which is exactly how `SparsePolynomialCommitment` starts on the wire (surge.rs:60-68); s, log_m, m follow as u64.
  /// `gens` is the reference's own argument (densified.rs:78-81): the caller's points, uploaded on first sight (HipProver::gens_on_device)
  pub fn commit<G: CurveGroup<ScalarField = F>>(&self, gens: &SparsePolyCommitmentGens<G>) -> SparsePolynomialCommitment<G> {
    let g = self.p.gens_on_device(gens, C, self.s, 0, self.log_m);
    let mut buf = call_bytes(|out, cap, len| unsafe { lasso_host_commit(self.d, g, out, cap, len) }, "lasso_host_commit");
    for v in [self.s as u64, self.log_m as u64, self.m as u64] { buf.extend_from_slice(&v.to_le_bytes()); }
    SparsePolynomialCommitment::<G>::deserialize_compressed(&buf[..]).expect("commitment bytes")
  }
}
impl<F: PrimeField, const C: usize> Drop for HipDensified<'_, F, C> {
  fn drop(&mut self) { unsafe { lasso_host_dense_free(self.d) } }
}

/// surge.rs:119-211 on the device, with the reference's own signature: the caller's live `&mut Transcript` and `&mut RandomTape<G>` (whatever they already
/// hold) are what the proof is bound to — the library appends to and draws from THEM through `lasso_host_prove_cb`, in the schedule of surge.rs:127-199,
/// so the proof bytes are those of the CPU prover given the same two objects (DESIGN.md §3 states what that claim rests on;
/// tests/test_transcript_callbacks_cpu.py holds the library to the oracle on pre-seeded transcripts).
pub fn prove_hip<G, const C: usize, const M: usize, S>(
  p: &HipProver,
  dense: &mut HipDensified<G::ScalarField, C>,
  r: &Vec<G::ScalarField>,
  gens: &SparsePolyCommitmentGens<G>,
  transcript: &mut Transcript,
  random_tape: &mut RandomTape<G>,
) -> SparsePolynomialEvaluationProof<G, C, M, S>
where
  G: CurveGroup,
  S: HipStrategy<G::ScalarField, C, M> + Sync,
  [(); S::NUM_SUBTABLES]: Sized,
  [(); S::NUM_MEMORIES]: Sized,
  [(); S::NUM_MEMORIES + 1]: Sized,
{
  assert_eq!(r.len(), ark_std::log2(dense.s) as usize); // surge.rs:131
  let st = S::descriptor();
  let g = p.gens_on_device(gens, C, dense.s, S::NUM_MEMORIES, dense.log_m); // the argument list of surge.rs:119-125, one to one: the caller's generators
  let (t_user, tape_user) = (transcript as *mut Transcript as *mut c_void, &mut random_tape.tape as *mut Transcript as *mut c_void);
  // A retry with a larger buffer would replay the protocol into transcripts that have already advanced, so the buffer is sized from the shape BEFORE the call
  // (ADVICE r4): a proof holds sqrt-sized point vectors for C + NUM_MEMORIES + 2C... commitments and openings plus O(log^2 s) scalars — bounded by
  // 32 bytes x (4 sqrt(next_pow2(2 C s)) + 4 sqrt(NUM_MEMORIES s)) + 64 KiB x (C + NUM_MEMORIES) with a wide margin (measured: 0.18 MB at C = 1, s = 2^24; 1.3 MB at C = 16, 2^24).
  let need = 32 * 8 * (((2 * C * dense.s) as f64).sqrt() as usize + ((S::NUM_MEMORIES * dense.s) as f64).sqrt() as usize + 2) + (1 << 16) * (C + S::NUM_MEMORIES + 2);
  let mut first = true;
  let bytes = call_bytes_sized(
    need.max(1 << 20),
    |out, cap, len| unsafe {
      assert!(first, "proof larger than the bound computed from its shape: a bug in that bound, not a condition to retry"); first = false;
      lasso_host_prove_cb(p.h, dense.d, g, &st, r.as_ptr() as *const lasso_fr, r.len(), &MERLIN_VTBL, t_user, &MERLIN_VTBL, tape_user, out, cap, len)
    },
    "lasso_host_prove_cb",
  );
  SparsePolynomialEvaluationProof::<G, C, M, S>::deserialize_compressed(&bytes[..]).expect("proof bytes")
}

/// calls that return bytes: -2 = buffer too small, *len = needed size
fn call_bytes(f: impl FnMut(*mut u8, usize, *mut usize) -> i32, what: &str) -> Vec<u8> { call_bytes_sized(1 << 24, f, what) }
fn call_bytes_sized(cap0: usize, mut f: impl FnMut(*mut u8, usize, *mut usize) -> i32, what: &str) -> Vec<u8> {
  let mut buf = vec![0u8; cap0]; // commitments: 32 bytes per row, 0.5 MB at 2^28 entries
  loop {
    let mut len = 0usize;
    let rc = f(buf.as_mut_ptr(), buf.len(), &mut len);
    if rc == -2 && len > buf.len() { buf.resize(len, 0); continue; }
    chk(rc, what);
    buf.truncate(len);
    return buf;
  }
}

// G = BN254 (BASELINE.json configs[1]): link liblasso_prover_bn254.so / liblasso_hip_bn254.so instead — same symbols, same code above
// (`G = ark_bn254::G1Projective`; the wire format is ark-ec's SWFlags encoding, which deserialize_compressed reads).
#[cfg(feature = "hip-bn254")]
const _: () = assert!(std::mem::size_of::<ark_bn254::Fr>() == 32 && std::mem::align_of::<ark_bn254::Fr>() == 8);

#[allow(dead_code)]
fn _uses(_: PolyCommitment<ark_curve25519::EdwardsProjective>) {}
