"""f-4, attribute-summary generation (reference graphs/createAttributeSum.py): the C MurmurHash3 x64-128 against
known answers -- the reference's OWN shipped files hold such hashes as node ids -- and the generator against the
shipped graphs/TEST/attr/{sum,map} files, byte for byte.  CPU only."""
import filecmp
import os

import pytest

from scaling_rgcn_training_amd import summaries as S
from tests.conftest import GOLDEN_DIR

TEST_DIR = os.path.join(GOLDEN_DIR, "TEST")


def test_murmur3_x64_128_known_answers():
    # ids that appear in the reference's graphs/TEST/attr/sum/*.nt: murmur3 of single-predicate sets
    assert S.hash128(b"<http://swrc.ontoware.org/ontology#isAbout>") == 66601471798836740805022011327225834224
    assert S.hash128(b"<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>") == 158298950357499570978480215079865772379
    # published MurmurHash3_x64_128 vectors (seed 0): empty input -> 0; the pangram's well-known digest
    # e34bbc7bbc071b6c7a433ca9c49a9347 is hex(h1) || hex(h2), and mmh3.hash128 returns h1 | h2 << 64
    assert S.hash128(b"") == 0
    assert S.hash128(b"The quick brown fox jumps over the lazy dog") == (0x7A433CA9C49A9347 << 64) | 0xE34BBC7BBC071B6C
    # every tail length 0..15 and a multi-block body differ from each other and are stable under the seed
    seen = {S.hash128(bytes(range(n))) for n in range(40)}
    assert len(seen) == 40
    assert S.hash128(b"abc", seed=1) != S.hash128(b"abc", seed=0)


def test_regenerating_the_shipped_test_summaries_is_byte_exact(tmp_path):
    """``create_sum_map(..., legacy=True)`` on TEST_complete.nt reproduces all six shipped files exactly (hash-named node
    ids included: the hash, the sorted-predicate-set key, the in + out sum and the dict-ordered map file all agree)."""
    sum_dir, map_dir = str(tmp_path / "sum") + os.sep, str(tmp_path / "map") + os.sep
    os.makedirs(sum_dir), os.makedirs(map_dir)
    S.create_sum_map(os.path.join(TEST_DIR, "TEST_complete.nt"), sum_dir, map_dir, "TEST", legacy=True)
    for kind in ("in", "out", "in_out"):
        assert filecmp.cmp(os.path.join(sum_dir, f"TEST_sum_{kind}.nt"), os.path.join(TEST_DIR, "attr", "sum", f"TEST_sum_{kind}.nt"),
                           shallow=False), kind
        assert filecmp.cmp(os.path.join(map_dir, f"TEST_map_{kind}.nt"), os.path.join(TEST_DIR, "attr", "map", f"TEST_map_{kind}.nt"),
                           shallow=False), kind


def test_current_reference_semantics_lowercase_and_type_excluded(tmp_path):
    """the script as it is in the reference today: terms lower-cased, rdf:type not part of the predicate sets"""
    lines = open(os.path.join(TEST_DIR, "TEST_complete.nt")).read().splitlines()
    out_h, in_h, both = S.property_hashes(lines)
    assert all(k == k.lower() for k in list(out_h) + list(in_h))
    about = "<http://swrc.ontoware.org/ontology#isabout>"
    pub = "<http://www.aifb.uni-karlsruhe.de/publikationen/viewpublikationowl/id1067instance>"
    assert out_h[pub] == S.hash128(about.encode())                       # its only non-type outgoing predicate
    assert "<http://swrc.ontoware.org/ontology#incollection>" not in in_h  # reached by rdf:type only
    assert S.LITERAL_KEY in in_h                                           # the literal object of line 8
    for e in both:
        assert both[e] == in_h.get(e, 0) + out_h.get(e, 0)
    sum_dir, map_dir = str(tmp_path) + os.sep, str(tmp_path) + os.sep
    S.create_sum_map(os.path.join(TEST_DIR, "TEST_complete.nt"), sum_dir, map_dir, "T")
    for kind in ("in", "out", "in_out"):
        s_lines = open(os.path.join(sum_dir, f"T_sum_{kind}.nt")).read().splitlines()
        assert len(s_lines) == len([l for l in lines if l.strip()])
        # the summary graph loads through the ingest path like any other graph
    from scaling_rgcn_training_amd import graphs as G
    g = G.Graph("sum")
    g.init_graph(G.parse_graph_nt(os.path.join(sum_dir, "T_sum_in_out.nt")))
    assert g.training_data.edge_index.shape[0] == 2 and g.num_nodes >= 2


def test_murmur3_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The host C code built with -fsanitize=address,undefined (CPU build only: no GPU sanitizer on this pool) and run over
    every tail length on exactly-sized heap buffers at odd offsets: an over-read by one byte or a misaligned wide load aborts
    the child.  Its digests must equal the product library's."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = os.path.join(os.path.dirname(S.HOST_LIB_PATH), "csrc", "murmur3_x64_128.c")
    main = tmp_path / "main.c"
    main.write_text(r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
void rgcn_murmur3_x64_128(const void* key, int64_t len, uint32_t seed, uint64_t out[2]);
int main(void) {
    for (int n = 0; n < 70; ++n)
        for (int off = 0; off < 3; ++off) {
            unsigned char* base = (unsigned char*)malloc((size_t)n + off + (n + off == 0));
            for (int i = 0; i < n; ++i) base[off + i] = (unsigned char)(i * 7 + 3);
            uint64_t out[2];
            rgcn_murmur3_x64_128(base + off, n, (uint32_t)off, out);
            printf("%d %d %016llx %016llx\n", n, off, (unsigned long long)out[0], (unsigned long long)out[1]);
            free(base);
        }
    return 0;
}
''')
    exe = tmp_path / "murmur_asan"
    subprocess.run([gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", str(main), src, "-o", str(exe)],
                   check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.split("\n")[:-1]
    assert len(lines) == 70 * 3
    for ln in lines:
        n, off, h1, h2 = ln.split()
        n, off = int(n), int(off)
        want = S.hash128(bytes((i * 7 + 3) & 255 for i in range(n)), seed=off)
        assert want == (int(h2, 16) << 64) | int(h1, 16), (n, off)
