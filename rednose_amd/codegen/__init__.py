"""HIP code generation for rednose_amd (model front end, expression lowering, kernel emitters)."""
