"""deepgroebner_amd — MI355X-native BuchbergerEnv step path (drop-in for deepgroebner's
LeadMonomialsEnv / CLeadMonomialsEnv and its ideal generators)."""
from .buchberger import (BuchbergerAgent, BuchbergerEnv, CLeadMonomialsEnv, LeadMonomialsAgent, LeadMonomialsEnv,  # noqa: F401
                         PolyLists, VecLeadMonomialsEnv, buchberger, interreduce, lead_monomials_vector, minimalize, reduce,
                         reduce_many, select, spoly, spoly_many, strategy_stats, update)
from .ideals import (FixedIdealGenerator, RandomBinomialIdealGenerator, RandomIdealGenerator,  # noqa: F401
                     basis, cyclic, degree_distribution, format_ideal, parse_ideal_dist, parse_ideal_string,
                     parse_polynomial)
