"""Twins of the reference's `utils` package: util_funcs, homophily_metrics (sparse flavour), homophily_plot (dense flavour)."""
