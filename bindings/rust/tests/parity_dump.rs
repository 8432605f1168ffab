// cargo test --test parity_dump -- --nocapture > dump.jsonl       (needs an MI355X for the `amd` lines)
//
// One JSON object per image and implementation, in the encoding of the repository's tests/golden/saddles_<image>.json:
// the reference crate itself ("impl":"crate") and this backend ("impl":"amd") on the reference's nine images.
//   python tools/compare_crate_dump.py dump.jsonl        compares the "crate" lines with the committed golden lists
// (tag ids and saddle counts identical, coordinates within 1e-3 px, k within 1e-4 relative, theta / phi within 1e-3 degrees;
// --strict: every bit).  AGX_REFERENCE_DIR = a checkout of powei-lin/aprilgrid-rs (its tests/data and data directories).
use image::ImageReader;

fn bits(v: f32) -> String {
    format!("\"{:08x}\"", v.to_bits())
}
fn list<T>(it: impl Iterator<Item = T>, f: impl Fn(T) -> String) -> String {
    format!("[{}]", it.map(f).collect::<Vec<_>>().join(","))
}
fn line(which: &str, image: &str, s: &[(f32, f32, f32, f32, f32)], tags: &mut Vec<(u32, [(f32, f32); 4])>) {
    tags.sort_by_key(|(id, _)| *id);
    println!(
        "{{\"impl\":\"{}\",\"image\":\"{}\",\"saddles\":{{\"x_bits\":{},\"y_bits\":{},\"k_bits\":{},\"theta_deg\":{},\"phi_deg\":{}}},\"tags\":{{{}}}}}",
        which,
        image,
        list(s.iter(), |p| bits(p.0)),
        list(s.iter(), |p| bits(p.1)),
        list(s.iter(), |p| bits(p.2)),
        list(s.iter(), |p| format!("{:.5}", p.3)),
        list(s.iter(), |p| format!("{:.5}", p.4)),
        tags.iter()
            .map(|(id, c)| format!("\"{}\":{}", id, list(c.iter(), |q| format!("[{},{}]", bits(q.0), bits(q.1)))))
            .collect::<Vec<_>>()
            .join(",")
    );
}

const FILES: [&str; 9] = [
    "tests/data/iphone.png", "tests/data/EuRoC.png", "tests/data/TUM_VI.png", "tests/data/right.png", "tests/data/r45.png",
    "tests/data/top.png", "tests/data/two_boards.png", "tests/data/top_right.png", "data/1520525725372653511.png",
];

#[test]
fn dump() {
    let root = std::env::var("AGX_REFERENCE_DIR").expect("AGX_REFERENCE_DIR = a checkout of powei-lin/aprilgrid-rs");
    let reference = aprilgrid::detector::TagDetector::new(&aprilgrid::TagFamily::T36H11, None);
    let amd = aprilgrid_amd::TagDetector::new(&aprilgrid_amd::TagFamily::T36H11, None);
    for path in FILES {
        let img = ImageReader::open(format!("{root}/{path}")).unwrap().decode().unwrap();
        let name = path.rsplit('/').next().unwrap();
        // the crate: src/detector.rs:408-446 and :505-540
        let s: Vec<_> = reference.refined_saddle_points(&img).iter().map(|p| (p.p.0, p.p.1, p.k, p.theta, p.phi)).collect();
        let mut tags: Vec<_> = reference.detect(&img).into_iter().collect();
        line("crate", name, &s, &mut tags);
        // this backend, same calls
        let s2: Vec<_> = amd.refined_saddle_points(&img).iter().map(|p| (p.p.0, p.p.1, p.k, p.theta, p.phi)).collect();
        let mut tags2: Vec<_> = amd.detect(&img).into_iter().collect();
        line("amd", name, &s2, &mut tags2);
        // what the reference's own tests assert (tests/test_detector.rs:21-32), for both
        assert_eq!(tags.len(), tags2.len(), "{name}: tag counts differ");
    }
}
