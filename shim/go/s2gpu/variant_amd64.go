//go:build amd64 && !appengine && !noasm && gc

package s2gpu

// The reference builds its assembly block encoders under exactly these constraints (s2/encode_amd64.go:1).
const defaultVariant = VariantAMD64
