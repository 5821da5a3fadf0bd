//! ref_pin -- regenerates the answers of tests/golden/*.npz with the REAL reference crate (hnsw_rs 0.3.4 + anndists 0.1).
//!
//! For every committed dump `<dir>/<name>.hnsw.{graph,data}` with a query file `<dir>/<name>.queries.bin` (written by
//! tests/golden/make_golden.py: header `u32 nq, u32 d, u32 k, u32 ef`, then nq*d little-endian f32) this program
//!   1. reloads the dump with `HnswIo::load_hnsw::<f32, D>()`             (src/hnswio.rs:431-524)
//!   2. answers the queries with `Hnsw::parallel_search(&queries, k, ef)`  (src/hnsw.rs:1612-1635)
//!      and, for `<name>.filter.bin` (sorted u64 ids), with `search_filter` (src/hnsw.rs:1487-1580)
//!   3. writes `<dir>/<name>.ref.bin`: `u32 nq, u32 k`, then per query `u32 count` and k records
//!      `{u64 d_id, u32 distance_bits, u8 layer, i32 rank}` (zero padded past count), all little endian.
//! Loading the COMMITTED dumps sidesteps the irreproducible StdRng level stream of construction: the graph is data,
//! the search on it is what gets pinned (ids, f32 distance bits, p_ids, counts, tie order of std's BinaryHeap and the
//! arithmetic of anndists included).
//!
//! The distance type is chosen from the file name prefix: l2_, l1_, cos_, dot_, hell_, jeff_, js_.
//!
//! Two probes localise a mismatch (inputs: tests/golden/make_pin_probes.py): `pin_pairs.bin` -> the crate's `eval` of fixed
//! vector pairs per distance and dimension (d = 1, 3, 25, 128, 784), and `pin_heap_scripts.bin` -> the pop order and
//! into_sorted_vec of std's BinaryHeap on tie-heavy scripts.  If the search fixtures differ, these say whether it is the
//! arithmetic (which distance, which dimension) or the heap order.
use std::fs;
use std::io::Write;
use std::path::Path;

use hnsw_rs::prelude::*; // re-exports hnswio::*, filter::* and anndists::dist::distances::* (src/prelude.rs)

fn read_u32(b: &[u8], off: usize) -> u32 {
    u32::from_le_bytes([b[off], b[off + 1], b[off + 2], b[off + 3]])
}

fn run<D>(dir: &Path, name: &str) -> anyhow::Result<()>
where
    D: Distance<f32> + Default + Send + Sync,
{
    let qbytes = fs::read(dir.join(format!("{name}.queries.bin")))?;
    let (nq, d, k, ef) = (
        read_u32(&qbytes, 0) as usize,
        read_u32(&qbytes, 4) as usize,
        read_u32(&qbytes, 8) as usize,
        read_u32(&qbytes, 12) as usize,
    );
    let mut queries: Vec<Vec<f32>> = Vec::with_capacity(nq);
    for i in 0..nq {
        let mut v = Vec::with_capacity(d);
        for j in 0..d {
            let off = 16 + 4 * (i * d + j);
            v.push(f32::from_le_bytes([qbytes[off], qbytes[off + 1], qbytes[off + 2], qbytes[off + 3]]));
        }
        queries.push(v);
    }
    let mut io = HnswIo::new(dir, name);
    let hnsw: Hnsw<f32, D> = io.load_hnsw::<f32, D>()?;
    let answers = hnsw.parallel_search(&queries, k, ef);
    let mut out = Vec::<u8>::new();
    out.extend_from_slice(&(nq as u32).to_le_bytes());
    out.extend_from_slice(&(k as u32).to_le_bytes());
    let mut emit = |out: &mut Vec<u8>, ans: &Vec<Neighbour>| {
        out.extend_from_slice(&(ans.len() as u32).to_le_bytes());
        for j in 0..k {
            if j < ans.len() {
                out.extend_from_slice(&(ans[j].d_id as u64).to_le_bytes());
                out.extend_from_slice(&ans[j].distance.to_bits().to_le_bytes());
                out.push(ans[j].p_id.0);
                out.extend_from_slice(&ans[j].p_id.1.to_le_bytes());
            } else {
                out.extend_from_slice(&[0u8; 17]);
            }
        }
    };
    for ans in &answers {
        emit(&mut out, ans);
    }
    fs::File::create(dir.join(format!("{name}.ref.bin")))?.write_all(&out)?;
    println!("{name}: {nq} queries, k={k}, ef={ef} -> {name}.ref.bin");
    // filtered search (sorted id vector = `impl FilterT for Vec<usize>`, src/filter.rs:11-15), if a filter file exists
    let fpath = dir.join(format!("{name}.filter.bin"));
    if fpath.exists() {
        let fb = fs::read(fpath)?;
        let filter: Vec<usize> = fb.chunks_exact(8).map(|c| u64::from_le_bytes(c.try_into().unwrap()) as usize).collect();
        let mut outf = Vec::<u8>::new();
        outf.extend_from_slice(&(nq as u32).to_le_bytes());
        outf.extend_from_slice(&(k as u32).to_le_bytes());
        for q in &queries {
            // the reference panics (unwrap on an emptied return_points) for some filters: recorded as count 0xFFFFFFFF
            let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| hnsw.search_filter(q, k, ef, Some(&filter))));
            match r {
                Ok(ans) => emit(&mut outf, &ans),
                Err(_) => {
                    outf.extend_from_slice(&u32::MAX.to_le_bytes());
                    outf.extend_from_slice(&vec![0u8; 17 * k]);
                }
            }
        }
        fs::File::create(dir.join(format!("{name}.ref_filter.bin")))?.write_all(&outf)?;
        println!("{name}: filtered search with {} allowed ids -> {name}.ref_filter.bin", filter.len());
    }
    Ok(())
}

// ---- probes that LOCALISE a mismatch (tests/golden/make_pin_probes.py wrote the inputs) ------------------------------------

/// `pin_pairs.bin`: blocks {u32 metric, u32 d, u32 n} + n x (a[d], b[d]) f32 -> `pin_pairs.ref.bin`: the same block headers,
/// each followed by n x u32 = `D::default().eval(a, b).to_bits()`.  Says WHICH distance at WHICH dimension differs.
fn probe_pairs(dir: &Path) -> anyhow::Result<()> {
    let path = dir.join("pin_pairs.bin");
    if !path.exists() {
        return Ok(());
    }
    let b = fs::read(path)?;
    let mut out = Vec::<u8>::new();
    let mut off = 0usize;
    while off + 12 <= b.len() {
        let (metric, d, n) = (read_u32(&b, off), read_u32(&b, off + 4) as usize, read_u32(&b, off + 8) as usize);
        off += 12;
        out.extend_from_slice(&metric.to_le_bytes());
        out.extend_from_slice(&(d as u32).to_le_bytes());
        out.extend_from_slice(&(n as u32).to_le_bytes());
        for _ in 0..n {
            let rd = |o: usize| -> Vec<f32> { (0..d).map(|j| f32::from_le_bytes([b[o + 4 * j], b[o + 4 * j + 1], b[o + 4 * j + 2], b[o + 4 * j + 3]])).collect() };
            let (va, vb) = (rd(off), rd(off + 4 * d));
            off += 8 * d;
            let v: f32 = match metric {
                0 => DistL2::default().eval(&va, &vb),
                1 => DistCosine::default().eval(&va, &vb),
                2 => DistDot::default().eval(&va, &vb),
                3 => DistL1::default().eval(&va, &vb),
                4 => DistHellinger::default().eval(&va, &vb),
                5 => DistJeffreys::default().eval(&va, &vb),
                _ => DistJensenShannon::default().eval(&va, &vb),
            };
            out.extend_from_slice(&v.to_bits().to_le_bytes());
        }
    }
    fs::File::create(dir.join("pin_pairs.ref.bin"))?.write_all(&out)?;
    println!("pin_pairs.bin -> pin_pairs.ref.bin");
    Ok(())
}

/// An entry ordered by its value ONLY, like `PointWithOrder` (src/hnsw.rs:283-297): the tag shows which of several equal
/// entries std's heap hands out.
#[derive(Clone, Copy)]
struct Tagged(f32, i32);
impl PartialEq for Tagged {
    fn eq(&self, o: &Self) -> bool {
        self.0 == o.0
    }
}
impl Eq for Tagged {}
impl PartialOrd for Tagged {
    fn partial_cmp(&self, o: &Self) -> Option<std::cmp::Ordering> {
        self.0.partial_cmp(&o.0)
    }
}
impl Ord for Tagged {
    fn cmp(&self, o: &Self) -> std::cmp::Ordering {
        self.partial_cmp(o).unwrap()
    }
}

/// `pin_heap_scripts.bin`: per script {u32 n_ops} + n_ops x {u8 is_pop, f32 value, i32 tag} -> `pin_heap_scripts.ref.bin`:
/// per script {u32 n_popped, tags..., u32 n_left, tags of into_sorted_vec...}.  Says whether std's BinaryHeap orders equal
/// entries the way the oracle's restatement does (push / pop / into_sorted_vec).
fn probe_heaps(dir: &Path) -> anyhow::Result<()> {
    let path = dir.join("pin_heap_scripts.bin");
    if !path.exists() {
        return Ok(());
    }
    let b = fs::read(path)?;
    let mut out = Vec::<u8>::new();
    let mut off = 0usize;
    while off + 4 <= b.len() {
        let n = read_u32(&b, off) as usize;
        off += 4;
        let mut heap = std::collections::BinaryHeap::<Tagged>::new();
        let mut popped = Vec::<i32>::new();
        for _ in 0..n {
            let is_pop = b[off] != 0;
            let val = f32::from_le_bytes([b[off + 1], b[off + 2], b[off + 3], b[off + 4]]);
            let tag = i32::from_le_bytes([b[off + 5], b[off + 6], b[off + 7], b[off + 8]]);
            off += 9;
            if is_pop {
                if let Some(e) = heap.pop() {
                    popped.push(e.1);
                }
            } else {
                heap.push(Tagged(val, tag));
            }
        }
        let left = heap.into_sorted_vec();
        out.extend_from_slice(&(popped.len() as u32).to_le_bytes());
        for t in &popped {
            out.extend_from_slice(&t.to_le_bytes());
        }
        out.extend_from_slice(&(left.len() as u32).to_le_bytes());
        for e in &left {
            out.extend_from_slice(&e.1.to_le_bytes());
        }
    }
    fs::File::create(dir.join("pin_heap_scripts.ref.bin"))?.write_all(&out)?;
    println!("pin_heap_scripts.bin -> pin_heap_scripts.ref.bin");
    Ok(())
}

fn main() -> anyhow::Result<()> {
    let dir = std::env::args().nth(1).unwrap_or_else(|| "tests/golden".to_string());
    let dir = Path::new(&dir);
    probe_pairs(dir)?;
    probe_heaps(dir)?;
    let mut names: Vec<String> = fs::read_dir(dir)?
        .filter_map(|e| e.ok())
        .filter_map(|e| e.file_name().to_str().and_then(|s| s.strip_suffix(".queries.bin").map(|s| s.to_string())))
        .collect();
    names.sort();
    for name in names {
        if name.starts_with("l2_") {
            run::<DistL2>(dir, &name)?;
        } else if name.starts_with("l1_") {
            run::<DistL1>(dir, &name)?;
        } else if name.starts_with("cos_") {
            run::<DistCosine>(dir, &name)?;
        } else if name.starts_with("dot_") {
            run::<DistDot>(dir, &name)?;
        } else if name.starts_with("hell_") {
            run::<DistHellinger>(dir, &name)?;
        } else if name.starts_with("jeff_") {
            run::<DistJeffreys>(dir, &name)?;
        } else if name.starts_with("js_") {
            run::<DistJensenShannon>(dir, &name)?;
        } else {
            eprintln!("skipping {name}: unknown distance prefix");
        }
    }
    Ok(())
}
