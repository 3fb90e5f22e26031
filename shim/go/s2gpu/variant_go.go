//go:build !(amd64 && !appengine && !noasm && gc)

package s2gpu

// Everywhere else the reference's block encoders are the portable Go ones (s2/encode_go.go:1).
const defaultVariant = VariantGo
