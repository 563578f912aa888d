"""CPU oracle for the ML+2PN inference hot path of wangxiaohit/GNNPN-SC.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gnnpn-sc_amd/`` (the product) may
import, call, link or execute anything in this package.  The only legitimate
importers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` — and there only as the checker / the timed CPU baseline,
never as the thing shipped.

What it is: a plain PyTorch-CPU (fp32) + numpy restatement of the reference's
algorithm for the path, each function citing the reference ``file:line`` it
follows.  It is pinned (see ``tests/golden/make_golden.py``) against outputs of
the *reference itself* imported in the build container:

* ``pn.py``    <- ``/root/reference/src/models/modelPN.py``  (real module, run on
  CPU with ``Tensor.cuda`` no-op'd)                          : PINNED
* ``data.py``  <- ``/root/reference/src/loadData.py``, ``src/ML2PN.py``
  (real modules, run on synthetic JSON datasets)             : PINNED
* ``ml.py``    <- ``/root/reference/src/models/modelML.py``: the reference's own
  ``Net.__init__``/``forward`` glue is run unmodified, but ``GINConv``,
  ``GCNConv`` (torch_geometric==1.7.0) and ``scatter`` (torch_scatter==2.0.6)
  are third-party, un-vendored and not installed in the image, so the conv
  arithmetic follows their published algorithm as restated in
  ``tests/golden/pyg_standin.py``.            : glue PINNED, conv "parity unpinned"
* ``woa.py``   <- ``/root/reference/src/baselines/WOA.py`` ``ESWOA`` (the
  step after the path; groundwork for a later round — no product code uses it yet): real class run
  with numpy's global generator routed to an explicit draw stream
  (``tests/golden/make_golden_woa.py``)                      : PINNED
"""
