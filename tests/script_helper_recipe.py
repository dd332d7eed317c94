"""Evaluation sequence for the update_best fixture: improvements, regressions and exact ties in either metric."""
EVALS = [(0, 0.40, 0.55), (20, 0.40, 0.55), (40, 0.40, 0.60), (60, 0.45, 0.58), (80, 0.45, 0.60), (100, 0.30, 0.60), (120, 0.45, 0.61),
         (140, 0.50, 0.50), (160, 0.50, 0.61), (180, 0.50, 0.61)]
