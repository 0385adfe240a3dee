"""CPU oracle for the sparse-MoE FFN block (TEST INFRASTRUCTURE ONLY).

Nothing in the product package (``moe_llava_amd``) may import from here.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.

PARITY UNPINNED: the algorithm restated here lives in the third-party package
``deepspeed==0.17.5`` (pinned by the reference in ``environment.yml:74``), which
is not vendored under the reference tree and is not installable offline.  The
reference tree holds no tests, golden vectors or fixtures for this path
(SURVEY.md section 4 / 8c).  The restatement therefore follows the published
upstream algorithm as transcribed in SURVEY.md section 8(a), and is pinned only by
(1) the hand-derived known-answer case of SURVEY.md section 8(a''), (2) the
bit-exact agreement of two independently written forms (dense-mask form in
``ds_moe_literal`` and index form in ``ds_moe_index``) and (3) algebraic
invariants of the routing.
"""
