"""CPU oracle for the hexgnn hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``gnn_hex_amd``) never
imports this package and fails loudly when its HIP library is missing.

Parity status
-------------
* model (``model_ref``): PARITY UNPINNED.  The arithmetic of the reference path
  lives in torch_geometric 2.2.0 / torch_scatter 2.1.0, which are absent from
  ``/root/reference`` and from this image, and the reference's own tests hold
  no golden vector for it.  The restatement follows the
  in-tree texts (``GN0/models.py``, ``GN0/torch_script_models.py``) line by
  line and is pinned only by hand-derived known-answer tests.
* env (``env_ref``): restates the python game logic with a canonical
  (ascending vertex id) iteration order; pinned by the closed-form start-graph
  sizes, the reference's winner-agreement property and the CSR container
  compiled from the reference's own ``graph.cpp`` (``oracle/_ref``).
"""
