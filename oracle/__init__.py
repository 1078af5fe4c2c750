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
be imported as-is.  ``tests/golden/make_goldens.py`` transliterates the reference
into ``/tmp`` (lib2to3 + integer-division fixes, nothing committed), imports
it in the build container and writes small input/output vectors to
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks this oracle
against every one of those vectors.  The image crop (``HumanAug.crop``) delegates
its pixels to scipy.misc (< 1.3) over PIL, both absent from the reference tree and
unpinned by it: the generator restates the three thin scipy wrappers over the REAL
Pillow of the build container (12.2) and runs the reference's own crop();
``oracle/crop.py`` restates the wrappers and Pillow's resampling arithmetic in numpy
and must reproduce those outputs byte for byte (it does) -- see DESIGN.md section 2.
Restatements WITHOUT a reference fixture (the reference's scripts cannot be imported
even transliterated): ``step.py`` -- the loop bodies of stack-hg.py,
joint-train-pose-s-r-agent.py (train_agent_sr) and the stage-2 scripts; they are
compositions of pinned pieces (model, losses, PCKh, gen_groundtruth) in the
reference's order, each line citing the script line it follows.
"""
