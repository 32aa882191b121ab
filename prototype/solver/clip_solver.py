"""Drop-in for `python -m prototype.solver.clip_solver --config ...` (reference solver/clip_solver.py):
the step runs on the MI355X HIP engine (declip_amd.solver)."""
from declip_amd.solver import ClsSolver, main  # noqa: F401

if __name__ == "__main__":
    main()
