"""ipk_amd -- MI355X-native phylo-k-mer scoring engine behind IPK's explore_kmers/explore_group seam.

The product is the C-ABI library ``libipkgpu.so`` (include/ipkgpu.h, ipk_amd/csrc/*.hip); this
package is the thin host mirror used by tests, bench.py and the build driver.  There is no CPU
fallback: without the HIP library or without a GPU every compute call raises.
"""
from .engine import Engine, IpkGpuError, Result, load_library, log_threshold, score_threshold, bits_per_symbol, kmer_batch, max_k  # noqa: F401
