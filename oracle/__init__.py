"""CPU oracle for the stacked-hourglass pose training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(``pose_adv_aug_amd``) may import, call or link anything in this directory;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and there only as the checker / the reported CPU
baseline, never as the thing that is measured or shipped.

What it is: a plain PyTorch-CPU fp32 / numpy restatement of the reference's
algorithm for every row of SURVEY.md section 8(a).  Each function cites the
reference file:line it follows (paths relative to the reference checkout).

How it is pinned: the reference has no tests, golden vectors or fixtures of
its own, and it is Python-2 / torch-0.3 source that cannot
be imported as-is.  ``tools/make_goldens.py`` transliterates the reference
into ``/tmp`` (lib2to3 + integer-division fixes, nothing committed), imports
it in the build container and writes small input/output vectors to
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks this oracle
against every one of those vectors.  The one exception is the image warp
(``HumanAug.crop``): its pixels depend on an unpinned scipy.misc/PIL pair, so
for that row only the *geometry* is pinned exactly and pixel parity is
"parity unpinned" (tolerance test against a PIL shim) -- see DESIGN.md.
"""
