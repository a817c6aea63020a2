"""
motifscan_amd -- MI355X-native drop-in for ONE hot path of shao-lab/MotifScan: the PWM scan
(`motifscan.scanner.Scanner.scan_motifs` -> `motifscan.motif.cscore.c_scan_motif`) and its sibling
`c_score`, behind the C-ABI in include/motifscan_amd.h.

    motifscan_amd.cscore    c_scan_motif / c_score with the reference's signatures
    motifscan_amd.scanner   Scanner / MotifSite / make_motif_sites / deduplicate_motif_sites
    motifscan_amd.matrix    PFM -> PPM -> PWM log-odds definitions
    motifscan_amd.dist      region sharding over the GPUs of a node + the one all-reduce
    motifscan_amd.synth     seeded synthetic workloads (bench.py, tests)

There is no CPU fallback anywhere in this package.
"""
__version__ = "0.1.0"
