for o in 2 1; do echo "== WDG_SELL_ORDER=$o"; WDG_SELL_ORDER=$o python scripts/dev/ab_quad_variants.py new; done
