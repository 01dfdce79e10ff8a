"""On-disk formats and input formatting either side of the ELBO path: a dependency-free MTZ
reader / writer, reciprocal-ASU bookkeeping from the symmetry operators in the file header, the mono / Laue formatters that
turn reflection tables into the `inputs` tuple, and the pre-formatted `.npz` container.  Host-side numpy: this is one-off
preprocessing, not the hot path."""
