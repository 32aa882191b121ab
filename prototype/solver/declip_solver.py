"""Drop-in for `python -m prototype.solver.declip_solver --config ...` (reference solver/declip_solver.py):
the step runs on the MI355X HIP engine (declip_amd.solver)."""
from declip_amd.solver import ClsSolver, main  # noqa: F401

if __name__ == "__main__":
    main()
