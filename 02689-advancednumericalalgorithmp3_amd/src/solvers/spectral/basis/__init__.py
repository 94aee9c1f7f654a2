from .spectral import (ChebyshevLobattoBasis, chebyshev_diff_matrix,  # noqa: F401
                       chebyshev_gauss_lobatto_nodes, clenshaw_curtis_weights)
