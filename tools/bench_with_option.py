"""Run bench.py with a diagnostic library option set first (A/B runs): python tools/bench_with_option.py <lstm_ablate value> [bench args].
lstm_ablate bit 6 (64) = the general cooperative decoder (decode_coop.hip) instead of the production build (decode_lean.hip)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
value = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench                                    # noqa: E402
from gnnpn_sc_amd import _lib                   # noqa: E402
_lib.check(_lib.load().gnnpn_set_option(b"lstm_ablate", value), "gnnpn_set_option")
bench.main()
