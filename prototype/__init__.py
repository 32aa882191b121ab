"""Drop-in `prototype` package: the reference's plug-in surface (prototype.model.model_entry,
prototype.loss_functions, prototype.utils.dist, prototype.solver) on the MI355X engine (declip_amd)."""
